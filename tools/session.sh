#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s57; mkdir -p $d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $d/pytest.txt
MKV_FUZZ_CASES=4000 MKV_FUZZ_CYCLE_CASES=300 MKV_FUZZ_NODEL_CASES=300 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $d/fuzz.txt
MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=400 MKV_FUZZ_CYCLE_CASES=2 MKV_FUZZ_NODEL_CASES=2 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $d/fuzz.txt
line() { python bench.py --no-cpu --no-pcie --no-abi "$@" 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], round(j['ms_per_step']*1000,1), j.get('repeats',{}).get('mops'))"; }
for ds in 1 0 1 0; do
 echo "== MKV_DEFER_SMALL=$ds"
 export MKV_DEFER_SMALL=$ds
 line --workload zipf50 --steps 20 --warmup 5
 line --workload zipf50 --steps 100 --warmup 5
 line --workload zipf50 --sets upd --steps 20 --warmup 5
 line --workload zipf50 --scheme 2choice --steps 20 --warmup 5
 line --steps 20 --warmup 5
 line --steps 20 --warmup 5 --lf 0.35
done 2>&1 | tee $d/lines.txt
