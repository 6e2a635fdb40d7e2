#!/bin/bash
# GPU round D: parity after the slow-path rewrite + a fuzz campaign + rates
set -u
out=gpurun_out/r2d; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest.log | cut -c1-300
( time timeout 1500 bash tools/fuzz_campaign.sh 1500 ) > $out/fuzz.log 2>&1; cat $out/fuzz.log
( MKV_FUZZ_BIG=1 MKV_FUZZ_CASES=600 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x -k fuzz_sequence 2>&1 | tail -2 ) | tee -a $out/fuzz.log
for chunk in 65536 131072; do
  echo "== LF 0.5 chunk $chunk" >> $out/probes.log
  MKV_INSERT_CHUNK=$chunk MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.5 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
done
echo "== LF 0.6 chunk 65536" >> $out/probes.log
MKV_INSERT_CHUNK=65536 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.6 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.85 52428" >> $out/probes.log
timeout 300 python tools/insert_probe.py 0.85 52428 6 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
cat $out/probes.log
