#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s34; mkdir -p $d
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $d/pytest.txt
MKV_FUZZ_CASES=400 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 | tee $d/fuzz.txt
line() { python bench.py --no-cpu --no-pcie --no-abi "$@" 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$1 $2 $3 $4 $5 $6', j['value'], j['ms_per_step']*1000, j.get('repeats',{}).get('mops'))"; }
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $d/drv.json; python -c "
import json; j=json.load(open('$d/drv.json')); print('drv', j['value'], j['repeats'], j['abi_three_call'], j['roofline'])"
line --steps 20 --warmup 5 --lf 0.5
line --steps 20 --warmup 5 --lf 0.85
line --steps 20 --warmup 5 --lf 0.2
line --steps 20 --warmup 5 --workload zipf50
line --steps 100 --warmup 5 --workload zipf50
line --steps 20 --warmup 5 --workload zipf95 --sets upd
line --steps 20 --warmup 5 --workload zipf50 --sets upd
