#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests/test_gpu_bench_cli.py tests/test_gpu_sharded.py -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
python bench.py --force-sharded --steps 40 --warmup 10 --no-cpu --no-pcie 2>gpurun_out/fs_err.txt | tail -1 > gpurun_out/fs_line.json
tail -5 gpurun_out/fs_err.txt
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/fs_line.json"))
print(d["value"], d["ms_per_step"], d["engine"], d["phases_us_per_rank"])
PY
