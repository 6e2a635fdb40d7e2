#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s47; mkdir -p $d
timeout 900 python -m pytest tests/test_gpu_abi.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 | tee $d/pytest.txt
for i in 1 2 3; do timeout 300 python -m pytest "tests/test_gpu_abi.py::test_cycle_workers_calls_on_alternating_streams_keep_issue_order" -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1; done
timeout 300 python tools/workers_rate.py 2>&1 | tail -6
