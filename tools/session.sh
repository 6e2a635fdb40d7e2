#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s50; mkdir -p $d
( time python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 ) 2>&1 | tail -6 | tee $d/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py --steps 20 --warmup 5 > $d/driver.json 2> $d/driver.err ) 2>&1 | grep real
python -c "
import json; j=json.load(open('$d/driver.json')); print(j['value'], j['ms_per_step'], j['repeats'], j['roofline']['frac'], j['roofline']['traffic_source'], j['cpu_baseline']['value'], j['abi_three_call']['value'], j['engine'])"
( time python bench.py > $d/default.json 2> $d/default.err ) 2>&1 | grep real
python -c "
import json; j=json.load(open('$d/default.json')); print(j['value'], j['ms_per_step'], j['repeats'], j['roofline']['frac'])"
ldd megakv_amd/_lib/libmegakv_hip.so | head -5
