#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s42; mkdir -p $d
{
echo "fuzz campaign on the round-4 tree (kernels of commit $(cat .head 2>/dev/null)): tools/fuzz_campaign.sh 10000 -- 10000 sequences + 24 + 24 cycle sequences + 11 forced-path cases per configuration"
bash tools/fuzz_campaign.sh 10000
echo "---- big tables (up to 2 MiB, batches up to 300 K requests): MKV_FUZZ_BIG=1, 600 sequences"
MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=600 MKV_FUZZ_CYCLE_CASES=2 MKV_FUZZ_NODEL_CASES=2 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1
echo "---- mkv_cycle sequences: 1500 with deletes, 1500 without (1 - 128 mark work-groups)"
MKV_FUZZ_CASES=2 MKV_FUZZ_CYCLE_CASES=1500 MKV_FUZZ_NODEL_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1
} 2>&1 | tee $d/campaign.txt
