#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s22; mkdir -p $d
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $d/pytest.txt 2>&1
grep -E "passed|failed|error" $d/pytest.txt | tail -3; grep -B5 -A25 "Error\|assert" $d/pytest.txt | head -60
exe=megakv_amd/_lib/scheduler_cycle
for wk in 1 4 16; do for mode in "0 0" "0 1" "1 0" "1 1"; do $exe 60 $wk $mode 2>&1 | tail -1; done; done | tee $d/scheduler.txt
