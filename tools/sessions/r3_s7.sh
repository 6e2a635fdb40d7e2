#!/bin/bash
set -u
out=gpurun_out/r3s7; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift
  timeout 600 python bench.py --no-cpu --no-pcie "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print(sys.argv[2], 'Mops', d['value'], 'us/step', round(d['ms_per_step']*1e3, 2), 'launch_us', r.get('launch_us'), 'host', round(d['host_submit_ms_per_step']*1e3,1), 'ok', d.get('get_loc_correct_last_timed'), d.get('set_loc_correct_last_timed'), d.get('table_entries_ok'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for rep in 1 2; do
run sh1_20_5_r$rep --force-sharded --steps 20 --warmup 5
run sh1_40_10_r$rep --force-sharded --steps 40 --warmup 10
run sh1_100_20_r$rep --force-sharded --steps 100 --warmup 20
run single_20_5_r$rep --steps 20 --warmup 5
done
