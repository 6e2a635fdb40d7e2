#!/bin/bash
# interleaved runs of one library under different environments (tunings) on ONE box:
#   tools/env_lines.sh <reps> "<bench args>" "<VAR=value ...|->" ["<VAR=value ...>" ...]     ("-": nothing set)
# prints value, us/step, search launch us per run
reps=$1; args=$2; shift 2
for rep in $(seq $reps); do
  for e in "$@"; do
    [ "$e" = "-" ] && set_env="" || set_env="$e"
    env $set_env timeout 600 python3 bench.py --no-cpu --no-pcie --no-live-traffic --no-abi --no-sched $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$e]', d['value'], round(d['ms_per_step']*1e3,2), d['roofline']['launch_us'])"
  done
done
