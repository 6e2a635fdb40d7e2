#!/bin/bash
# Runs on the GPU box: the bench lines behind DESIGN.md's results table -> gpurun_out/table/*.json
mkdir -p gpurun_out/table
run() { name=$1; shift; timeout 900 python3 bench.py --no-pcie "$@" > gpurun_out/table/$name.json 2> gpurun_out/table/$name.err; python3 - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/table/{n}.json"))
    r = d.get("roofline") or {}
    print(f"{n:16s} {d['value']:10.1f} Mops {d['ms_per_step']*1e3:9.1f} us/step  search {r.get('avg_launch_us')} us frac {r.get('frac')} hit {d.get('get_hit_rate')} cpu {(d.get('cpu_baseline') or {}).get('value')}")
except Exception as e:
    print(n, "FAILED", e)
PY
}
run default
run threecall --memset --overlap none --no-cpu
run get100 --workload get100 --no-cpu
run keyref --keying ref --no-cpu
run z50_20 --workload zipf50 --steps 20 --warmup 5 --no-cpu
run z50_60 --workload zipf50 --steps 60 --warmup 5 --no-cpu
run z50_100 --workload zipf50 --steps 100 --warmup 5 --no-cpu
run z50_2choice --workload zipf50 --scheme 2choice --steps 20 --warmup 5 --no-cpu
run lf035 --lf 0.35 --no-cpu
run lf05 --lf 0.5 --no-cpu
run lf085 --lf 0.85 --steps 20 --no-cpu
run supd --sets upd --steps 20 --warmup 5 --no-cpu
run sharded1 --force-sharded --no-cpu
