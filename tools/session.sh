#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s49; mkdir -p $d
line() { python bench.py --no-cpu --no-pcie --no-abi "$@" 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], round(j['ms_per_step']*1000,1), j.get('repeats',{}).get('mops'))"; }
for m in 1 2 1 2; do
 echo "== MKV_MERGE_ROUND2=$m"
 export MKV_MERGE_ROUND2=$m
 line --steps 20 --warmup 5 --lf 0.5
 line --steps 20 --warmup 5 --lf 0.35
 line --steps 40 --warmup 5 --lf 0.85
done 2>&1 | tee $d/merge.txt
