#!/bin/bash
bash tools/sessions/r3_s14.sh 600 2>&1 | tail -24 | grep -v "rc=0"
B="python bench.py --no-cpu --no-pcie --steps 20 --warmup 5 --update-info --no-events"
for w in "--workload zipf95 --sets upd" "--workload zipf50 --sets upd" "--workload zipf50" "--workload zipf95" "--workload zipf95 --lf 0.5" "--workload zipf50 --steps 100"; do
    echo "== $w"
    timeout 300 $B $w 2>&1 | grep -E "update info|\"value\"" | cut -c1-400
done
echo "== default 100/20"; for i in 1 2; do timeout 300 python bench.py --no-cpu --no-pcie 2>/dev/null | cut -c1-260; done
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fuzz.py 2>&1 | grep -E "passed|failed" | tail -3
