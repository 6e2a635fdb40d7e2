#!/bin/bash
out=gpurun_out/ab; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in new old; do
  arg=""; [ $v = old ] && arg="_ab/libmegakv_hip_r01.so"
  python3 tools/probes/empty_insert.py $arg
  rm -rf $out/$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -- python3 tools/probes/empty_insert.py $arg > /dev/null 2>&1
  f=$(ls $out/$v/*/*kernel_stats.csv | head -1); echo "== $v"; head -6 $f | cut -d, -f1-7 | cut -c1-40,150-400
done
