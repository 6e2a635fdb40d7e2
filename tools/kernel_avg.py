#!/usr/bin/env python3
"""Average duration per kernel over the LAST n bench steps of a rocprofv3 kernel trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_search" in r["Kernel_Name"]]
sel = rows[marks[-n]:]
agg = collections.defaultdict(list)
for r in sel:
    name = r["Kernel_Name"].replace("void ", "").replace("mkv::", "").split("(")[0][:40]
    agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
print("  ".join(f"{k}={sum(v)/len(v):.1f}" for k, v in agg.items()))
