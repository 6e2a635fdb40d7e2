#!/bin/bash
cd "$GRAFT_REPO_ROOT"
ab() { echo "== $*"; bash tools/ab_lines.sh 2 "--no-abi $*" r04 r05a tree; }
ab --workload zipf50 --steps 20 --warmup 5
ab --workload zipf50 --steps 100 --warmup 5
ab --workload zipf50 --scheme 2choice --steps 20 --warmup 5
