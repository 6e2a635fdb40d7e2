#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s45; mkdir -p $d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $d/pytest.txt
bash tools/collect_profiles.sh driver --steps 20 --warmup 5 2>&1 | tail -2 | cut -c1-200
bash tools/collect_profiles.sh default 2>&1 | tail -2 | cut -c1-200
PMC=0 bash tools/collect_profiles.sh zipf50 --workload zipf50 --steps 20 --warmup 5 2>&1 | tail -2 | cut -c1-200
PMC=0 bash tools/collect_profiles.sh supd --sets upd --steps 20 --warmup 5 2>&1 | tail -2 | cut -c1-200
PMC=0 bash tools/collect_profiles.sh zipf50upd --workload zipf50 --sets upd --steps 20 --warmup 5 2>&1 | tail -2 | cut -c1-200
PMC=0 bash tools/collect_profiles.sh sharded1 --force-sharded 2>&1 | tail -2 | cut -c1-200
bash tools/results_table.sh > $d/table.log 2>&1; tail -52 gpurun_out/table/table.txt | cut -c1-200
