#!/bin/bash
# interleaved A/B of library builds on ONE box:  tools/ab_lines.sh <reps> "<bench args>" <name> [<name> ...]
# (name "tree": the tree's library; else _ab/libmegakv_hip_<name>.so); prints value, us/step, launch us per run
reps=$1; args=$2; shift 2
lib=megakv_amd/_lib/libmegakv_hip.so
cp $lib /tmp/lib_tree.so
for rep in $(seq $reps); do
  for v in "$@"; do
    [ $v = tree ] && cp /tmp/lib_tree.so $lib || cp _ab/libmegakv_hip_$v.so $lib
    old=""; case $v in r0*) old=1;; esac
    MKV_LIB_OLD_BUILD=$old timeout 600 python3 bench.py --no-cpu --no-pcie --no-live-traffic --no-abi --no-sched $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], round(d['ms_per_step']*1e3,2), d['roofline']['launch_us'])"
  done
done
cp /tmp/lib_tree.so $lib
