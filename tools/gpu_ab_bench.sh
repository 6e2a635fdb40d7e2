#!/bin/bash
# A/B of two builds of the library on ONE box:  tools/gpu_ab_bench.sh <other .so> [bench args]
# (runs bench.py alternately with the tree's library and with the other one in its place)
other=$1; shift
lib=megakv_amd/_lib/libmegakv_hip.so
cp $lib /tmp/lib_a.so
for rep in 1 2 3; do
  for v in a b; do
    [ $v = a ] && cp /tmp/lib_a.so $lib || cp $other $lib
    timeout 600 python3 bench.py --no-cpu --no-pcie "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], round(d['ms_per_step']*1e3,2), d['roofline']['launch_us'])"
  done
done
cp /tmp/lib_a.so $lib
