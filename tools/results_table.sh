#!/bin/bash
# Runs on the GPU box: the bench lines behind DESIGN.md's results table -> gpurun_out/table/*.json,
# the table itself -> gpurun_out/table/table.txt (copied to profiles/<round>_results_table.txt)
mkdir -p gpurun_out/table
exec > >(tee gpurun_out/table/table.txt) 2>&1
run() { name=$1; shift; timeout 900 python3 bench.py "$@" > gpurun_out/table/$name.json 2> gpurun_out/table/$name.err; python3 - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/table/{n}.json"))
    r = d.get("roofline") or {}
    p = d.get("pcie_inclusive") or {}
    c = d.get("cpu_baseline") or {}
    print(f"{n:14s} {d['value']:9.1f} Mops {d['ms_per_step']*1e3:8.1f} us/step  launch {r.get('launch_us')} us (p10 {r.get('p10_us')} p90 {r.get('p90_us')}) frac {r.get('frac')} "
          f"get_ok {d.get('get_loc_correct_last_timed')} set_ok {d.get('set_loc_correct_last_timed')} entries_ok {d.get('table_entries_ok')} "
          f"step_GBps {(d.get('step_algorithmic') or {}).get('GBps')} cpu {c.get('value')} cpu_mt {(c.get('all_threads') or {}).get('value')} "
          f"pcie {p.get('value')} ({p.get('us_per_step')} us) one_stream {(p.get('one_stream') or {}).get('value')}")
except Exception as e:
    print(n, "FAILED", e)
PY
}
run default
run driver --steps 20 --warmup 5
run threecall --memset --overlap none --no-cpu --no-pcie
run get100 --workload get100 --no-pcie
run keyref --keying ref --no-cpu --no-pcie
run z50_20 --workload zipf50 --steps 20 --warmup 5 --no-cpu --no-pcie
run z50_60 --workload zipf50 --steps 60 --warmup 5 --no-cpu --no-pcie
run z50_100 --workload zipf50 --steps 100 --warmup 5 --no-cpu --no-pcie
run z50_2choice --workload zipf50 --scheme 2choice --steps 20 --warmup 5 --no-cpu --no-pcie
run z50_2c_100 --workload zipf50 --scheme 2choice --steps 100 --warmup 5 --no-cpu --no-pcie
run lf035 --lf 0.35 --no-cpu --no-pcie
run lf05 --lf 0.5 --no-cpu --no-pcie
run lf085 --lf 0.85 --steps 40 --no-cpu --no-pcie
run supd --sets upd --steps 20 --warmup 5 --no-cpu --no-pcie
run supd_2c --sets upd --scheme 2choice --steps 20 --warmup 5 --no-cpu --no-pcie
run z50_upd --workload zipf50 --sets upd --steps 10 --warmup 3 --no-cpu --no-pcie
run z50_upd_2c --workload zipf50 --sets upd --scheme 2choice --steps 10 --warmup 3 --no-cpu --no-pcie
run sharded1 --force-sharded --no-cpu
echo "-- insert only (tools/insert_probe.py: 2^19 new keys per call)"
for lf in 0.2 0.35 0.5 0.6; do echo "LF $lf"; timeout 300 python3 tools/insert_probe.py $lf 524288 4 2>&1 | grep batch | cut -c1-230; done
echo "-- probes (tools/probes): host-buffer copies, two-stream cycle"
make -s -C tools/probes pcie_probe overlap_probe > /dev/null 2>&1
timeout 300 ./tools/probes/pcie_probe 2>&1 | tee gpurun_out/table/pcie_probe.txt
timeout 120 ./tools/probes/overlap_probe 2>&1 | tee gpurun_out/table/overlap_probe.txt
