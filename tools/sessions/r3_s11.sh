#!/bin/bash
set -u
out=gpurun_out/r3s11; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/t -- ./megakv_amd/_lib/graph_cycle 30 20 50 30 > $out/line.txt 2> $out/err.txt
cat $out/line.txt
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
k = sorted(csv.DictReader(open(glob.glob(out + '/t/*/*kernel_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(k) if 'k_search_mark' in r['Kernel_Name']]
for lo in (marks[20], marks[60]):
    prev_end = None
    for r in k[lo - 1:lo + 8]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(r['Kernel_Name'][:30], 'dur', (e - s) / 1000, 'gap', (s - prev_end) / 1000 if prev_end else 0)
        prev_end = e
    print()
PY
