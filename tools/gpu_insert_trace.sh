#!/bin/bash
# kernel timeline of the last insert call of tools/insert_probe.py:  tools/gpu_insert_trace.sh <lf> [batch] [new|upd] [tuning=value ...]
lf=$1; batch=${2:-524288}; kind=${3:-new}
out=gpurun_out/itrace_$lf; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/insert_probe.py $lf $batch 3 $kind cuckoo "${@:4}" > $out/out.txt 2> $out/err.txt
grep batch $out/out.txt | cut -c1-330
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tails = [i for i, r in enumerate(rows) if "k_tail" in r["Kernel_Name"]]
lo = tails[-1] - 3
prev = None
for r in rows[lo:tails[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("mkv::", "").split("(")[0][:34]
    print(f"{n:34s} dur {(e - s) / 1000:7.1f}  gap {((s - prev) / 1000 if prev else 0):6.1f}  grid {r['Grid_Size_X']}")
    prev = e
PY
