#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s60; mkdir -p $d
{
echo "fuzz campaign on the round's LAST kernels ($(cat .head)): tools/fuzz_campaign.sh 6000 -- 6000 sequences + 24 + 24 cycle sequences + 13 forced-path cases per configuration"
bash tools/fuzz_campaign.sh 6000
echo "---- rounds on 2 048 work-groups (shared list places), 6000 sequences"
MKV_FUZZ_ROUND_BLOCKS=2048 MKV_FUZZ_CASES=6000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1
echo "---- big tables: MKV_FUZZ_BIG=1, 600 sequences"
MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=600 MKV_FUZZ_CYCLE_CASES=2 MKV_FUZZ_NODEL_CASES=2 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1
echo "---- mkv_cycle sequences: 1000 with deletes, 1000 without"
MKV_FUZZ_CASES=2 MKV_FUZZ_CYCLE_CASES=1000 MKV_FUZZ_NODEL_CASES=1000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1
} 2>&1 | tee $d/campaign.txt
