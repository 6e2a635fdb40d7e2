#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s26; mkdir -p $d
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $d/pytest.txt 2>&1
grep -E "passed|failed|error" $d/pytest.txt | tail -3
( MKV_FUZZ_CASES=600 MKV_FUZZ_CYCLE_CASES=200 MKV_FUZZ_NODEL_CASES=200 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8 ) > $d/fuzz.txt 2>&1
grep -E "passed|failed|error" $d/fuzz.txt | tail -3
( MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=150 MKV_FUZZ_CYCLE_CASES=1 MKV_FUZZ_NODEL_CASES=1 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q -k test_fuzz_sequence 2>&1 | tail -8 ) > $d/fuzzbig.txt 2>&1
grep -E "passed|failed|error" $d/fuzzbig.txt | tail -3
run() { name=$1; shift; env "$@" > $d/$name.json 2> $d/$name.err; \
	python - $d/$name.json $name <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], x["value"], x["ms_per_step"], x["repeats"]["us_per_step"])
except Exception as e:
    print(sys.argv[2], "bad", e)
PY
grep "update info" $d/$name.err | sed 's/.*round1/round1/' | cut -c1-330; }
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --no-abi --update-info"
run drv $B
run z50 $B --workload zipf50
run z50_100 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu --no-pcie --no-abi --workload zipf50
run z50upd $B --workload zipf50 --sets upd
run supd $B --sets upd
run lf05 $B --lf 0.5
run z50_2c $B --workload zipf50 --scheme 2choice
bash tools/gpu_trace.sh z50 --workload zipf50 --steps 20 --warmup 5 --no-abi --repeats 1 > $d/trace_z50.txt 2>&1; head -12 $d/trace_z50.txt
