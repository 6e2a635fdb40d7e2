#!/bin/bash
set -u
out=gpurun_out/r3s10; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $out/t -- python3 bench.py --no-cpu --no-pcie --no-events --steps 20 --warmup 5 > $out/line.json 2> $out/err.txt
ls $out/t/*/
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
k = sorted(csv.DictReader(open(glob.glob(out + '/t/*/*kernel_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
api = sorted(csv.DictReader(open(glob.glob(out + '/t/*/*hip_api_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
print(api[0].keys())
by_corr = {r['Correlation_Id']: r for r in api}
marks = [i for i, r in enumerate(k) if 'k_search_mark' in r['Kernel_Name']]
lo = marks[len(marks) // 2]
prev_end = None
for r in k[lo:lo + 10]:
    a = by_corr.get(r['Correlation_Id'])
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(r['Kernel_Name'][:30], 'start', s % 10**9 / 1000, 'dur', (e - s) / 1000, 'gap', (s - prev_end) / 1000 if prev_end else 0,
          'api', a['Function'] if a else None, 'api_start-before-kernel-start', (s - int(a['Start_Timestamp'])) / 1000 if a else None)
    prev_end = e
PY
