#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s53; mkdir -p $d
line() { python bench.py --no-cpu --no-pcie --no-abi "$@" 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], round(j['ms_per_step']*1000,1), j.get('repeats',{}).get('mops'))"; }
for tr in 48 0 1 48 0 1; do
 echo "== MKV_TAIL_ROUNDS=$tr"
 export MKV_TAIL_ROUNDS=$tr
 line --workload zipf50 --lf 0.5 --steps 10 --warmup 3
 line --workload zipf50 --steps 100 --warmup 5
 line --steps 40 --warmup 5 --lf 0.85
 line --steps 20 --warmup 5 --lf 0.5
 line --workload zipf50 --sets upd --steps 20 --warmup 5
done 2>&1 | tee $d/tail_rounds.txt
