#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -k "chain_hint" 2>&1 | tail -15
