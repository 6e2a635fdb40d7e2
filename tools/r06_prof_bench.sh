#!/bin/bash
# kernel trace of a bench.py command, the timed region's steps printed launch by launch:
#   tools/r06_prof_bench.sh <tag> [bench flags...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-live-traffic --no-cpu --no-pcie --no-abi --no-sched --repeats 1 "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/b_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_search_mark' in r['Kernel_Name'] or 'k_search_part' in r['Kernel_Name'] or ('k_search<' in r['Kernel_Name'])]
start=idx[-8]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:start+14]:
    n=r['Kernel_Name']
    n=n[n.find('mkv::')+5:][:34] if 'mkv::' in n else n[:34]
    print(f"{n:36s} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
PY
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $OUT/bench.log | head -2
