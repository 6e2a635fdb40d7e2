#!/bin/bash
bash tools/sessions/r3_s14.sh 400 2>&1 | tail -24 | grep -v "rc=0"
B="python bench.py --no-cpu --no-pcie --steps 10 --warmup 3 --update-info --no-events"
for e in "MKV_DUP_GROUPS=1" "MKV_DUP_GROUPS=1 MKV_LAZY_ALT=0"; do
  for w in "--workload zipf95 --sets upd" "--workload zipf50 --sets upd" "--workload zipf50 --sets upd --scheme 2choice" "--workload zipf50"; do
    echo "== $e $w"
    env $e timeout 300 $B $w 2>&1 | grep -E "update info|\"value\"" | cut -c1-560
  done
done
