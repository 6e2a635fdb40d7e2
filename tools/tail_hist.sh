#!/bin/bash
# per-kernel duration statistics of the steps of a bench run (rocprofv3 kernel trace): tools/tail_hist.sh <name> [bench args]
name=$1; shift
out=gpurun_out/trace_$name; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu --no-pcie --kernel-launches 8 "$@" > $out/line.json 2> $out/err.txt
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("mkv::", "").split("(")[0].split("<")[0]
    if n.startswith("k_"):
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for n, v in d.items():
    v2 = sorted(v)
    print(f"{n:22s} n {len(v):4d}  min {v2[0]:6.1f}  p50 {v2[len(v)//2]:6.1f}  mean {sum(v)/len(v):6.1f}  max {v2[-1]:6.1f}")
t = d.get("k_tail", [])
if t:
    print("k_tail > 6 us:", sum(1 for x in t if x > 6), "of", len(t), " durations:", " ".join(f"{x:.1f}" for x in t[-40:]))
PY
