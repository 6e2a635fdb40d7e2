#!/bin/bash
# kernel trace of bench.py's scheduler_regime block (16 workers x 7000/6000/1000, mkv_cycle_workers)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_regime
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o rg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-live-traffic --no-cpu --no-pcie --no-abi --repeats 1 > $OUT/bench.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/rg_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_search_multi_mark' in r['Kernel_Name']]
# the enqueued mode: cycles 4..47 of the first pass -> take around the 30th launch
start=idx[30]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:start+13]:
    n=r['Kernel_Name']
    n=n[n.find('mkv::')+5:][:34] if 'mkv::' in n else n[:34]
    print(f"{n:36s} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
PY
