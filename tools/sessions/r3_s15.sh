#!/bin/bash
# groups: whole GPU suite, the probes, SET-updates workloads with and without groups, a longer campaign
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fuzz.py 2>&1 | tail -5
timeout 120 python tests/debug/streams_probe.py 2>&1 | tail -3
timeout 120 python tests/debug/dup_probe.py 2>&1 | tail -6
B="python bench.py --no-cpu --no-pcie --steps 20 --warmup 5"
for dg in 1 0; do
  for w in "--workload zipf95 --sets upd" "--workload zipf50 --sets upd" "--workload zipf50 --sets upd --scheme 2choice" "--workload zipf95" "--workload zipf50"; do
    echo -n "dup_groups=$dg $w: "
    MKV_DUP_GROUPS=$dg timeout 300 $B $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,2),'Gops', round(d['ms_per_step']*1e3,1),'us')"
  done
done
echo -n "100 steps zipf50 upd: "; timeout 300 python bench.py --no-cpu --no-pcie --steps 100 --warmup 20 --workload zipf50 --sets upd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,2),'Gops', round(d['ms_per_step']*1e3,1),'us')"
MKV_FUZZ_CASES=2000 bash tools/sessions/r3_s14.sh 2000 2>&1 | tail -24
