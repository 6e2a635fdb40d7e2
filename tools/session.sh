#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export MKV_FUZZ_CYCLE_CASES=400 MKV_FUZZ_NODEL_CASES=400
bash tools/fuzz_campaign.sh 6000 2>&1 | tee gpurun_out/fuzz_r05.txt
echo "== big tables"; MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=600 MKV_FUZZ_CYCLE_CASES=24 MKV_FUZZ_NODEL_CASES=24 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1 | tee -a gpurun_out/fuzz_r05.txt
echo "== cycles"; MKV_FUZZ_CASES=8 MKV_FUZZ_CYCLE_CASES=3000 MKV_FUZZ_NODEL_CASES=3000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1 | tee -a gpurun_out/fuzz_r05.txt
