#!/bin/bash
# round 3: one-rank sharded step with the new pack; timelines of the two schedules
set -u
out=gpurun_out/r3s4; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | tail -3
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
run sharded1 --force-sharded --steps 60 --warmup 10
MKV_SHARD_FUSED=1 run sharded1_fused --force-sharded --steps 60 --warmup 10
MKV_SHARD_TRACE=4 run sharded1_trace --force-sharded --steps 60 --warmup 10
grep -E "lane [0-9]" $out/sharded1_trace.err | tail -40
