#!/bin/bash
set -u
out=gpurun_out/r2g; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q --timeout=600 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest.log | cut -c1-300
for mode in 0 1; do MKV_FUZZ_CASES=1500 MKV_FUZZ_DEBUG_SKIP=$mode timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -x -k fuzz_sequence 2>&1 | tail -1; done
MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=500 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x -k fuzz_sequence 2>&1 | tail -1
echo "== LF 0.5" >> $out/probes.log
timeout 300 python tools/insert_probe.py 0.5 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.6" >> $out/probes.log
timeout 300 python tools/insert_probe.py 0.6 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.85 52428" >> $out/probes.log
timeout 300 python tools/insert_probe.py 0.85 52428 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
cat $out/probes.log
for a in "" "--lf 0.5" "--lf 0.85 --steps 40" "--workload zipf50 --steps 100 --warmup 5" "--workload zipf50 --steps 25 --warmup 5 --lf 0.2" "--workload zipf50 --steps 25 --warmup 5 --lf 0.3" "--workload zipf50 --steps 25 --warmup 5 --lf 0.4" "--workload zipf50 --steps 25 --warmup 5 --lf 0.5" "--workload zipf50 --steps 25 --warmup 5 --lf 0.55" "--workload zipf50 --scheme 2choice --steps 25 --warmup 5"; do
  timeout 600 python bench.py --no-cpu --no-pcie --kernel-launches 16 --update-info $a 2>$out/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$a', d['value'], d['ms_per_step'], r['launch_us'], d['get_loc_correct_last_timed'], d['set_loc_correct_last_timed'], d['table_entries'])"
  grep "update info" $out/err.txt | cut -c1-420
done
