#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s18; mkdir -p $d
run() { name=$1; shift; env "$@" > $d/$name.json 2> $d/$name.err; \
	python - $d/$name.json $name <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], x["value"], x["ms_per_step"], x.get("roofline", {}).get("launch_us"), x.get("set_loc_correct_last_timed"), x.get("table_entries_ok"))
except Exception as e:
    print(sys.argv[2], "bad", e)
PY
grep "update info" $d/$name.err | sed 's/.*round2/round2/' | cut -c1-250; }
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --update-info"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $d/pytest.txt 2>&1
grep -E "passed|failed|error" $d/pytest.txt | tail -3
( MKV_FUZZ_CASES=600 MKV_FUZZ_CYCLE_CASES=200 MKV_FUZZ_NODEL_CASES=200 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8 ) > $d/fuzz.txt 2>&1
grep -E "passed|failed|error" $d/fuzz.txt | tail -3
run new1 $B
run old1 MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B
run new2 $B
run old2 MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B
run new_def timeout 300 python bench.py --no-cpu --no-pcie
run old_def MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 timeout 300 python bench.py --no-cpu --no-pcie
run new_z50 $B --workload zipf50
run old_z50 MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B --workload zipf50
run new_supd $B --sets upd
run old_supd MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B --sets upd
run new_z50upd $B --workload zipf50 --sets upd
run old_z50upd MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B --workload zipf50 --sets upd
run new_lf05 $B --lf 0.5
run old_lf05 MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B --lf 0.5
run new_sh $B --force-sharded
run old_sh MKV_SELF_HELP=0 MKV_MERGE_ROUND2=0 $B --force-sharded
run new_2c $B --scheme 2choice
run new_ref $B --keying ref
