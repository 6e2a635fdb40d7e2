#!/bin/bash
# GPU round C: insert-path experiments (pass size, load factor)
set -u
out=gpurun_out/r2c; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
for chunk in 65536 131072 262144; do
  echo "== LF 0.5 chunk $chunk" >> $out/probes.log
  MKV_INSERT_CHUNK=$chunk MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.5 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
done
for lf in 0.2 0.35; do
  echo "== LF $lf default chunking" >> $out/probes.log
  timeout 300 python tools/insert_probe.py $lf 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
done
echo "== LF 0.6 chunk 65536" >> $out/probes.log
MKV_INSERT_CHUNK=65536 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.6 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
echo "== LF 0.6 chunk 131072" >> $out/probes.log
MKV_INSERT_CHUNK=131072 MKV_CALM_FACTOR=1 timeout 300 python tools/insert_probe.py 0.6 524288 4 2>&1 | grep batch | cut -c1-330 >> $out/probes.log
cat $out/probes.log
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p $out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/insert_probe.py 0.5 524288 6 > $out/prof.log 2>&1
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-200
