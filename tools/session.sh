#!/bin/bash
# the GPU session of the moment (rewritten per gpurun call); see tools/README.md
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 | tail -1
