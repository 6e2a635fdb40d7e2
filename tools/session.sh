#!/bin/bash
# One GPU session (gpurun -- bash tools/session.sh): rewritten for every call; what it printed is kept
# under gpurun_out/.
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s13; mkdir -p $d
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $d/pytest.txt 2>&1
grep -E "passed|failed|error" $d/pytest.txt | tail -3
run() { name=$1; shift; env "$@" > $d/$name.json 2> $d/$name.err; \
	python - $d/$name.json $name <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], x["value"], x["ms_per_step"], x.get("roofline", {}).get("launch_us"), x.get("set_loc_correct_last_timed"), x.get("table_entries_ok"))
except Exception as e:
    print(sys.argv[2], "bad", e)
PY
}
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie"
run drv1 $B
run drv2 $B
run def1 timeout 300 python bench.py --no-cpu --no-pcie
run supd $B --sets upd
run z50 $B --workload zipf50
run z50upd $B --workload zipf50 --sets upd
run lf05 $B --lf 0.5
run sh1 $B --force-sharded
run get100 $B --workload get100
bash tools/gpu_trace.sh base --steps 20 --warmup 5 > $d/trace.txt 2>&1; head -9 $d/trace.txt
timeout 200 python tools/loopback_trace.py 8 2>&1 | tail -5
