#!/bin/bash
set -u
out=gpurun_out/r3s13; mkdir -p $out
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
for lz in 3 2 1; do
export MKV_LAZY_ALT=$lz
run default_lz$lz --steps 20 --warmup 5
run z50_20_lz$lz --workload zipf50 --steps 20 --warmup 5
run z50_100_lz$lz --workload zipf50 --steps 100 --warmup 5
run lf05_lz$lz --lf 0.5 --steps 20 --warmup 5
run lf085_lz$lz --lf 0.85 --steps 40 --warmup 5
done
