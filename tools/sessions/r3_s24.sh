#!/bin/bash
# round 3, final tree: profiles, timelines, results table, fuzz campaign
bash tools/sessions/r3_s22.sh > gpurun_out/s24_profiles.log 2>&1
bash tools/results_table.sh > gpurun_out/s24_table.log 2>&1
( echo "fuzz campaign on the round's final tree: $(date -u +%F) -- tools/sessions/r3_s14.sh 2000 (2000 sequences + 24 cycle sequences + 11 forced-path cases per configuration)"; bash tools/sessions/r3_s14.sh 2000 2>&1 | grep -v "rc=0" ) > gpurun_out/s24_fuzz.log 2>&1
( echo "big tables (MKV_FUZZ_BIG=1)"; MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1 ) >> gpurun_out/s24_fuzz.log 2>&1
tail -3 gpurun_out/s24_fuzz.log
