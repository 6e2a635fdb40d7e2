#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
ab() { echo "== $*"; bash tools/ab_lines.sh 3 "--no-abi $*" r04 tree; }
ab --sets upd --steps 20 --warmup 5
ab --workload zipf50 --sets upd --steps 10 --warmup 3
