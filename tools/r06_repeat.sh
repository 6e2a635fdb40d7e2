#!/bin/bash
# the driver's command N times in a row on one box (profiles/r06_repeat.txt): one line per process
n=${1:-6}
for i in $(seq $n); do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=d.get('abi_three_call') or {}; s=d.get('scheduler_regime') or {}
print('run $i: %.1f Mops %.2f us/step  launch %s us  frac %s  traffic %s  three-call %s  deferred %s  cycle_workers enqueued %s Mops  cpu %s' % (
  d['value'], d['ms_per_step']*1e3, r.get('launch_us'), r.get('frac'), r.get('traffic'), a.get('value'), (a.get('deferred') or {}).get('value') if isinstance(a.get('deferred'), dict) else a.get('deferred_value'),
  s.get('mops_enqueued'), (d.get('cpu_baseline') or {}).get('value')))"
done
