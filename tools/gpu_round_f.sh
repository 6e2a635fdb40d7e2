#!/bin/bash
set -u
out=gpurun_out/r2f; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
python3 tools/probes/empty_insert.py; python3 tools/probes/empty_insert.py _ab/libmegakv_hip_r01.so
( time timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q --timeout=600 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest.log | cut -c1-300
for mode in 0 1 2; do MKV_FUZZ_CASES=1200 MKV_FUZZ_DEBUG_SKIP=$mode timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -x -k fuzz_sequence 2>&1 | tail -1; done
echo "== LF 0.5 chunk 65536" >> $out/probes.log
MKV_INSERT_CHUNK=65536 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.5 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.5 chunk 131072" >> $out/probes.log
MKV_INSERT_CHUNK=131072 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.5 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.6 chunk 65536" >> $out/probes.log
MKV_INSERT_CHUNK=65536 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.6 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.85 52428" >> $out/probes.log
timeout 300 python tools/insert_probe.py 0.85 52428 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
cat $out/probes.log
for a in "" "--lf 0.5" "--lf 0.85 --steps 40" "--workload zipf50 --steps 100 --warmup 5"; do
  timeout 600 python bench.py --no-cpu --no-pcie --kernel-launches 16 $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$a', d['value'], d['ms_per_step'], r['launch_us'], d['get_loc_correct_last_timed'], d['set_loc_correct_last_timed'], d['table_entries'])"
done
