#!/bin/bash
# round 3, session 1: this round's box baseline + the sorted-GET upper bound
set -u
out=gpurun_out/r3s1; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift
  timeout 600 python bench.py --no-cpu --no-pcie "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print(sys.argv[2], 'Mops', d['value'], 'us/step', round(d['ms_per_step']*1e3, 2), 'launch_us', r.get('launch_us'), 'frac', r.get('frac'), 'ok', d.get('get_loc_correct_last_timed'), d.get('set_loc_correct_last_timed'), d.get('table_entries_ok'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run default --steps 20 --warmup 5
run default_sorted --steps 20 --warmup 5 --sorted-gets
run get100 --workload get100 --steps 20 --warmup 5
run get100_sorted --workload get100 --steps 20 --warmup 5 --sorted-gets
run default100 
run sharded1 --force-sharded --steps 40 --warmup 10
run upd95 --sets upd --steps 20 --warmup 5
run upd50 --workload zipf50 --sets upd --steps 10 --warmup 3
run z50 --workload zipf50 --steps 20 --warmup 5
