#!/usr/bin/env python3
"""Prints the kernel timeline of the last N insert calls (or bench steps) from a
rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous
kernel.  usage: trace_timeline.py <kernel_trace.csv> [steps]"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("mkv::", "")
    return n.split("(")[0][:46]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "k_search" in r["Kernel_Name"]]
    for start in marks[-steps - 1:-1]:
        nxt = marks[marks.index(start) + 1]
        t0 = int(rows[start]["Start_Timestamp"])
        prev_end = None
        for r in rows[start:nxt]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (s - prev_end) / 1000 if prev_end else 0.0
            print(f"{(s - t0) / 1000:9.1f}us  dur {(e - s) / 1000:7.1f}  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
            prev_end = e
        print(f"   step span {(int(rows[nxt]['Start_Timestamp']) - t0) / 1000:.1f} us\n")


if __name__ == "__main__":
    main()
