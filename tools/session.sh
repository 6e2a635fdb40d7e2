#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
