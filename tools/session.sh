#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s41; mkdir -p $d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $d/pytest.txt
MKV_FUZZ_CASES=600 MKV_FUZZ_CYCLE_CASES=150 MKV_FUZZ_NODEL_CASES=100 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $d/fuzz.txt
line() { python bench.py --no-cpu --no-pcie --no-abi "$@" 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], round(j['ms_per_step']*1000,1), j.get('repeats',{}).get('mops'))"; }
for tree in . _ab/r4a . _ab/r4a; do
 echo "== tree $tree"
 ( cd $tree
 line --steps 20 --warmup 5
 line --steps 20 --warmup 5 --sets upd
 line --steps 20 --warmup 5 --workload zipf50
 line --steps 40 --warmup 5 --lf 0.85 )
done 2>&1 | tee $d/lines.txt
for a in "zipf95 upd 6" "zipf50 new 6"; do echo "== tail phases $a"; timeout 300 python tools/tail_phases.py $a 2>&1 | grep "^step" | cut -c1-400; done | tee $d/tail_phases.txt
