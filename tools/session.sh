#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export MKV_FUZZ_CYCLE_CASES=300 MKV_FUZZ_NODEL_CASES=300
bash tools/fuzz_campaign.sh 2000 2>&1 | tee gpurun_out/fuzz_r05_s2.txt | tail -18
echo "-- big tables"
MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=300 MKV_FUZZ_CYCLE_CASES=24 MKV_FUZZ_NODEL_CASES=24 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1 | tee -a gpurun_out/fuzz_r05_s2.txt
