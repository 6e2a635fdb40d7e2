#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests/test_gpu_abi.py tests/test_gpu_reference_drivers.py -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head
python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie | tail -1 > gpurun_out/s_line.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/s_line.json"))
print(d["value"], d["ms_per_step"], d.get("abi_three_call"))
PY
