#!/usr/bin/env python3
"""Kernel timeline between the last two occurrences of a marker kernel.
usage: trace_window.py <kernel_trace.csv> <marker-substring>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1000 if prev else 0.0
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1000:8.1f} dur {(e - s) / 1000:6.1f} gap {gap:6.1f} {name}")
    prev = e
print("window", (int(rows[b]["Start_Timestamp"]) - t0) / 1000, "us")
