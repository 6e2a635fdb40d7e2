#!/bin/bash
# interleaved A/B of library builds on ONE box, the scheduler_regime block of the bench line:
#   tools/ab_sched.sh <reps> <name> [<name> ...]      (name "tree": the tree's library; else _ab/libmegakv_hip_<name>.so)
reps=$1; shift
lib=megakv_amd/_lib/libmegakv_hip.so
cp $lib /tmp/lib_tree.so
for rep in $(seq $reps); do
  for v in "$@"; do
    [ $v = tree ] && cp /tmp/lib_tree.so $lib || cp _ab/libmegakv_hip_$v.so $lib
    timeout 600 python3 bench.py --no-cpu --no-pcie --no-live-traffic --no-abi --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['scheduler_regime']; print('$v', 'enqueued', s['us_per_cycle_enqueued'], 'us', s['mops_enqueued'], 'Mops; synced', s['us_per_cycle_synced'], 'us; correct', s['searches_correct'])"
  done
done
cp /tmp/lib_tree.so $lib
