#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s59; mkdir -p $d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 | tee $d/pytest.txt
MKV_FUZZ_CASES=3000 MKV_FUZZ_CYCLE_CASES=100 MKV_FUZZ_NODEL_CASES=100 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $d/fuzz.txt
MKV_FUZZ_ROUND_BLOCKS=2048 MKV_FUZZ_CASES=3000 MKV_FUZZ_CYCLE_CASES=100 MKV_FUZZ_NODEL_CASES=100 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $d/fuzz.txt
MKV_FUZZ_ROUND_BLOCKS=1024 MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=300 MKV_FUZZ_CYCLE_CASES=2 MKV_FUZZ_NODEL_CASES=2 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $d/fuzz.txt
python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie 2>/dev/null | tail -1 | cut -c1-200
