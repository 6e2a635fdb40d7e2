#!/usr/bin/env python3
"""All kernels of the last few steps of a rocprofv3 --kernel-trace CSV with their queue
(stream) ids, sorted by start time.  usage: trace_gantt.py <kernel_trace.csv> <marker> [n markers]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
a, b = marks[-n - 2], marks[-2]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
for r in rows[a:b]:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("mkv::", "").split("(")[0][:28]
    print(f"{(s - t0) / 1000:8.1f} -> {(e - t0) / 1000:8.1f}  q{q}  {'    ' * q}{name} ({(e - s) / 1000:.1f})")
print("span", (int(rows[b]["Start_Timestamp"]) - t0) / 1000, "us for", n, "steps")
