#!/bin/bash
# SET-updates workloads: what the insert side does (update info), by feature
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fuzz.py 2>&1 | grep -E "passed|failed" | tail -3
timeout 120 python tests/debug/streams_probe.py 4 2>&1 | tail -2
B="python bench.py --no-cpu --no-pcie --steps 10 --warmup 3 --update-info --no-events"
for e in "MKV_DUP_GROUPS=1" "MKV_DUP_GROUPS=0" "MKV_DUP_GROUPS=0 MKV_LAZY_ALT=0"; do
  for w in "--workload zipf95 --sets upd" "--workload zipf50 --sets upd"; do
    echo "== $e $w"
    env $e timeout 300 $B $w 2>&1 | grep -E "update info|\"value\"" | cut -c1-900
  done
done
