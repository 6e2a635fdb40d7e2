#!/bin/bash
# kernel timeline of the last steps of a bench run:  tools/gpu_trace.sh <name> [bench args]
name=$1; shift
out=gpurun_out/trace_$name; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu --no-pcie --kernel-launches 8 "$@" > $out/line.json 2> $out/err.txt
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_search_mark" in r["Kernel_Name"] or "k_search_slots_mark" in r["Kernel_Name"]]
if not marks:       # three-call sequence: a step starts with the memset of the results (or the search)
    marks = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "k_search" in r["Kernel_Name"]]
# steps from the middle of the run (the timed region, if it is longer than warm-up + the launch-timing pass)
lo = marks[len(marks) // 2]
prev = None
for r in rows[lo:lo + 22]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("mkv::", "").split("(")[0][:34]
    print(f"{n:34s} dur {(e - s) / 1000:7.1f}  gap {((s - prev) / 1000 if prev else 0):6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev = e
PY
