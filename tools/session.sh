#!/bin/bash
# the GPU session of the moment (rewritten per gpurun call): the round's last check of the committed tree
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s58; mkdir -p $d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $d/pytest.txt
MKV_FUZZ_CASES=5000 MKV_FUZZ_CYCLE_CASES=500 MKV_FUZZ_NODEL_CASES=500 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $d/fuzz.txt
MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=500 MKV_FUZZ_CYCLE_CASES=2 MKV_FUZZ_NODEL_CASES=2 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee -a $d/fuzz.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/collect_profiles.sh driver --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-200
bash tools/collect_profiles.sh default 2>&1 | tail -1 | cut -c1-200
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('driver command run $i: value', j['value'], 'repeats', j['repeats']['mops'], 'abi_three_call', j['abi_three_call']['value'], 'launch_us', j['roofline']['launch_us'], j['roofline']['traffic_source'])"; done | tee $d/driver_runs.txt
