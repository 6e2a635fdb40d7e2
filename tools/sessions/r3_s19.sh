#!/bin/bash
# per-kernel time of the SET-updates workloads
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for w in "zipf95" "zipf50"; do
  raw=gpurun_out/s19_$w; rm -rf $raw; mkdir -p $raw
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $raw -- python3 bench.py --no-cpu --no-pcie --no-events --steps 10 --warmup 3 --workload $w --sets upd > $raw.json 2> $raw.err
  echo "== $w upd rc=$?"
  f=$(find $raw -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 $f | cut -c1-220
done
