#!/bin/bash
# round 3: default sharded schedule, repeated and interleaved (lanes 2/3/4), and with RCCL's own copy (SELF=0)
set -u
out=gpurun_out/r3s6; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_sharded.py -q 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -8
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
for rep in 1 2 3; do
for L in 2 3 4; do
MKV_SHARD_LANES=$L run sh1_L${L}_r$rep --force-sharded --steps 100 --warmup 20
done
done
MKV_SHARD_SELF=0 MKV_SHARD_LANES=2 run sh1_a2a_L2 --force-sharded --steps 100 --warmup 20
MKV_SHARD_SELF=0 MKV_SHARD_LANES=3 MKV_SHARD_SVC=2 run sh1_a2a_L3_svc2 --force-sharded --steps 100 --warmup 20
MKV_SHARD_SELF=0 MKV_SHARD_LANES=4 MKV_SHARD_SVC=2 run sh1_a2a_L4_svc2 --force-sharded --steps 100 --warmup 20
run sh1_driver --force-sharded --steps 20 --warmup 5
