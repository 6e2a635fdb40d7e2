#!/bin/bash
set -u
out=gpurun_out/r3s9; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
run() { name=$1; shift
  timeout 600 python bench.py --no-cpu --no-pcie "$@" > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print(sys.argv[2], 'Mops', d['value'], 'us/step', round(d['ms_per_step']*1e3, 2), 'launch_us', r.get('launch_us'), 'ok', d.get('get_loc_correct_last_timed'), d.get('set_loc_correct_last_timed'), d.get('table_entries_ok'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for rep in 1 2; do
run default_r$rep --steps 20 --warmup 5
done
run default100
run sharded1 --force-sharded --steps 20 --warmup 5
bash tools/gpu_trace.sh default --steps 20 --warmup 5 2>&1 | tail -14
