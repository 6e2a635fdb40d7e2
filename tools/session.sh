#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/ab_lines.sh 2 "--no-abi --workload zipf50 --steps 20 --warmup 5" tree fullbk
bash tools/ab_lines.sh 2 "--no-abi --steps 20 --warmup 5" tree fullbk
cp _ab/libmegakv_hip_fullbk.so megakv_amd/_lib/libmegakv_hip.so
bash tools/gpu_trace.sh z50 --no-abi --workload zipf50 --steps 20 --warmup 5 | head -9
