#!/bin/bash
set -u
out=gpurun_out/r2h; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
for chunk in 131072 262144 524288; do
for lf in 0.3 0.4 0.5 0.55; do
  MKV_INSERT_CHUNK=$chunk MKV_CALM_FACTOR=1 timeout 600 python bench.py --no-cpu --no-pcie --kernel-launches 8 --update-info --workload zipf50 --steps 25 --warmup 5 --lf $lf 2>$out/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('chunk $chunk lf $lf', d['value'], d['ms_per_step'])"
  grep "update info" $out/err.txt | cut -c150-520
done; done
