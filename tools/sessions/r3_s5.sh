#!/bin/bash
# round 3: one-rank sharded step: self-slot bypass, schedules
set -u
out=gpurun_out/r3s5; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_sharded.py -q 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -8
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
run sharded1_default --force-sharded --steps 60 --warmup 10
MKV_SHARD_SELF=0 run sharded1_a2a --force-sharded --steps 60 --warmup 10
MKV_SHARD_SELF=0 MKV_SHARD_SVC=2 run sharded1_a2a_svc2 --force-sharded --steps 60 --warmup 10
MKV_SHARD_LANES=2 run sharded1_l2 --force-sharded --steps 60 --warmup 10
MKV_SHARD_LANES=4 run sharded1_l4 --force-sharded --steps 60 --warmup 10
MKV_SHARD_SVC=2 run sharded1_svc2 --force-sharded --steps 60 --warmup 10
MKV_SHARD_PACED=0 run sharded1_unpaced --force-sharded --steps 60 --warmup 10
MKV_SHARD_SCHED=plain run sharded1_plain --force-sharded --steps 60 --warmup 10
MKV_SHARD_TRACE=4 run sharded1_trace --force-sharded --steps 60 --warmup 10
grep -E "lane [0-9]" $out/sharded1_trace.err | tail -24
