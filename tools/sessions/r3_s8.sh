#!/bin/bash
# round 3: profiles of the one-rank sharded step (kernel stats + timeline), loopback-8 timeline, CLI test with 8 ranks
set -u
out=gpurun_out/r3s8; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_bench_cli.py -q 2>&1 | tail -3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sharded1 -- python3 bench.py --no-cpu --no-pcie --force-sharded --steps 20 --warmup 5 > $out/sharded1.json 2> $out/sharded1.err
echo "rocprof rc=$?"
f=$(ls $out/sharded1/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-160
MKV_SHARD_TRACE=3 python bench.py --no-cpu --no-pcie --no-events --force-sharded --steps 60 --warmup 10 > $out/trace1.json 2> $out/trace1.err
grep -E "lane [0-9]" $out/trace1.err | tail -16 > $out/trace1.txt; cat $out/trace1.txt
