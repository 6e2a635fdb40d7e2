#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_abi.py -x -q 2>&1 | tail -4
for rep in 1 2 3; do for p in 1 2; do
MKV_PLANS=$p timeout 600 python3 bench.py --no-cpu --no-pcie --no-abi --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('plans $p', d['value'], round(d['ms_per_step']*1e3,2), d['roofline']['launch_us'], d['set_loc_correct_last_timed'], d['table_entries_ok'])"
done; done
bash tools/gpu_trace.sh default --no-abi --steps 20 --warmup 5 | head -12
