#!/bin/bash
# GPU round E: parity subset, probes, kernel timelines
set -u
out=gpurun_out/r2e; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q --timeout=600 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest.log | cut -c1-300
MKV_FUZZ_CASES=1500 MKV_FUZZ_DEBUG_SKIP=1 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -x -k fuzz_sequence 2>&1 | tail -1
echo "== LF 0.5 chunk 65536" >> $out/probes.log
MKV_INSERT_CHUNK=65536 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.5 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.6 chunk 65536" >> $out/probes.log
MKV_INSERT_CHUNK=65536 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.6 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.85 52428" >> $out/probes.log
timeout 300 python tools/insert_probe.py 0.85 52428 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
cat $out/probes.log
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for cfg in "default:" "lf50:--lf 0.5" "z50:--workload zipf50 --steps 40"; do
  name=${cfg%%:*}; args=${cfg#*:}
  mkdir -p $out/prof_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$name -- python3 bench.py --no-cpu --no-pcie --kernel-launches 8 $args > $out/prof_$name.json 2> $out/prof_$name.err
  f=$(ls $out/prof_$name/*/*kernel_trace.csv | head -1)
  echo "---- $name"; python tools/trace_timeline.py $f 2 | tail -40
done
