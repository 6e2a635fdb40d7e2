#!/bin/bash
# round 3 evidence: profiles of the final tree
set -u
bash tools/collect_profiles.sh default
PMC=0 bash tools/collect_profiles.sh driver --steps 20 --warmup 5
bash tools/collect_profiles.sh zipf50 --workload zipf50 --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh zipf50upd --workload zipf50 --sets upd --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh supd --sets upd --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh sharded1 --force-sharded --steps 20 --warmup 5
echo "== step timeline (default)"; bash tools/gpu_trace.sh default --no-events | tee gpurun_out/r3_step_timeline.txt
echo "== step timeline (sharded, one rank)"; bash tools/gpu_trace.sh sharded1 --force-sharded --no-events --steps 20 --warmup 5 | tee gpurun_out/r3_sharded1_timeline.txt
echo "== loopback 8"; MKV_SHARD_TRACE=3 timeout 600 python tools/loopback_trace.py 8 64 > gpurun_out/r3_loopback8.txt 2> gpurun_out/r3_loopback8.err; cat gpurun_out/r3_loopback8.txt; grep -E "lane|batch" gpurun_out/r3_loopback8.err | tail -40
echo "== loopback 1"; MKV_SHARD_TRACE=3 timeout 600 python tools/loopback_trace.py 1 64 > gpurun_out/r3_loopback1.txt 2> gpurun_out/r3_loopback1.err; cat gpurun_out/r3_loopback1.txt
echo "== sorted GETs"
for a in "" "--sorted-gets" "--workload get100" "--workload get100 --sorted-gets"; do echo -n "[$a] "; timeout 300 python bench.py --no-cpu --no-pcie --steps 20 --warmup 5 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'Mops', round(d['ms_per_step']*1e3,1),'us/step; launch', r['launch_us'], 'us frac', r['frac'])"; done | tee gpurun_out/r3_sorted_gets.txt
echo "== event probe"; make -s -C tools/probes event_probe && timeout 120 ./tools/probes/event_probe 2>&1 | tee gpurun_out/r3_event_probe.txt
