#!/bin/bash
# the GPU session of the moment (rewritten per gpurun call): a long soak of the seeded fuzz test on the product library
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s52; mkdir -p $d
{
echo "soak on the round-4 final kernels ($(cat .head)): the default configuration, seeds 0 .. 119 999 in four processes of 30 000 (MKV_FUZZ_FIRST), + big tables 2 000"
for first in 0 30000 60000 90000; do
  echo -n "seeds $first .. $((first+29999)): "; MKV_FUZZ_FIRST=$first MKV_FUZZ_CASES=30000 MKV_FUZZ_CYCLE_CASES=2 MKV_FUZZ_NODEL_CASES=2 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x -k "test_fuzz_sequence" 2>&1 | grep -E "passed|failed" | tail -1
done
echo -n "big tables, seeds 0 .. 1999: "; MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=2000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x -k "test_fuzz_sequence" 2>&1 | grep -E "passed|failed" | tail -1
} 2>&1 | tee $d/soak.txt
