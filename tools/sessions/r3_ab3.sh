#!/bin/bash
# several builds of the library on one box, alternating: r3_ab3.sh "<bench args>" <so> <so> ...
args=$1; shift
lib=megakv_amd/_lib/libmegakv_hip.so
cp $lib /tmp/lib_head.so
for rep in 1 2; do
  for so in /tmp/lib_head.so "$@"; do
    cp $so $lib
    echo -n "$so: "
    timeout 300 python3 bench.py --no-cpu --no-pcie --kernel-launches 8 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], round(d['ms_per_step']*1e3,2))"
  done
done
cp /tmp/lib_head.so $lib
