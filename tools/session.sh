#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s36; mkdir -p $d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $d/pytest.txt
MKV_FUZZ_CASES=300 MKV_FUZZ_CYCLE_CASES=100 MKV_FUZZ_NODEL_CASES=100 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $d/fuzz.txt
bash tools/collect_profiles.sh driver --steps 20 --warmup 5 2>&1 | tail -6
bash tools/collect_profiles.sh default 2>&1 | tail -6
bash tools/collect_profiles.sh get100 --workload get100 --prewarm-ms 0 --no-abi 2>&1 | tail -6
PMC=0 bash tools/collect_profiles.sh zipf50 --workload zipf50 --steps 20 --warmup 5 2>&1 | tail -3
PMC=0 bash tools/collect_profiles.sh supd --sets upd --steps 20 --warmup 5 2>&1 | tail -3
PMC=0 bash tools/collect_profiles.sh zipf50upd --workload zipf50 --sets upd --steps 20 --warmup 5 2>&1 | tail -3
