#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_abi.py -x -q 2>&1 | tail -2
ab() { echo "== $*"; bash tools/ab_lines.sh 2 "--no-abi $*" r04 r05a tree; }
ab --workload zipf50 --steps 20 --warmup 5
ab --lf 0.5 --steps 20 --warmup 5
ab --lf 0.35 --steps 20 --warmup 5
ab --workload zipf50 --steps 100 --warmup 5
ab --steps 20 --warmup 5
