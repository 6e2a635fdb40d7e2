#!/bin/bash
# final tree after the tail's last change: profiles of the driver's and the default command again,
# the Zipf-SET lines of the results table, whole GPU suite
bash tools/collect_profiles.sh driver --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh default
PMC=0 bash tools/collect_profiles.sh zipf50upd --workload zipf50 --sets upd --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh supd --sets upd --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh zipf50 --workload zipf50 --steps 20 --warmup 5
for a in "--workload zipf50 --sets upd --steps 20 --warmup 5" "--workload zipf50 --sets upd --scheme 2choice --steps 20 --warmup 5" "--workload zipf50 --sets upd --steps 10 --warmup 3" "--sets upd --steps 20 --warmup 5" "--sets upd --scheme 2choice --steps 20 --warmup 5" "--workload zipf50 --steps 20 --warmup 5" "--workload zipf50 --steps 100 --warmup 5" "--steps 20 --warmup 5" ""; do echo -n "[$a] "; timeout 300 python bench.py --no-cpu --no-pcie $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'Mops', round(d['ms_per_step']*1e3,1),'us/step; launch', r['launch_us'], 'us frac', r['frac'], 'set_ok', d.get('set_loc_correct_last_timed'), 'entries_ok', d.get('table_entries_ok'))"; done | tee gpurun_out/r3_supd_lines.txt
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
