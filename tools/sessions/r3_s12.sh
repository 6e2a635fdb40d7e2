#!/bin/bash
set -u
out=gpurun_out/r3s12; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift
  timeout 900 python bench.py --no-cpu --no-pcie "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print(sys.argv[2], 'Mops', d['value'], 'us/step', round(d['ms_per_step']*1e3, 2), 'launch_us', r.get('launch_us'), 'ok', d.get('get_loc_correct_last_timed'), d.get('set_loc_correct_last_timed'), d.get('table_entries_ok'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run default --steps 20 --warmup 5
run z50_20 --workload zipf50 --steps 20 --warmup 5
run z50_60 --workload zipf50 --steps 60 --warmup 5
run z50_100 --workload zipf50 --steps 100 --warmup 5
run z50_2c --workload zipf50 --scheme 2choice --steps 20 --warmup 5
run lf035 --lf 0.35 --steps 20 --warmup 5
run lf05 --lf 0.5 --steps 20 --warmup 5
run lf085 --lf 0.85 --steps 40 --warmup 5
run upd95 --sets upd --steps 20 --warmup 5
run upd50 --workload zipf50 --sets upd --steps 6 --warmup 2
for lf in 0.2 0.5 0.6; do python tools/insert_probe.py $lf 524288 6 2>&1 | tail -2; done
