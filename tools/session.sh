#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_abi.py -m gpu -q -W default 2>&1 | tail -15
