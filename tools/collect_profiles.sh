#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace/stats of the default
# bench command plus separate PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit counters,
# each in its own run, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# Raw output -> gpurun_out/profiles_raw/<tag>/ ; tools/make_profiles.py turns it
# into the committed summaries under profiles/.
# usage: tools/collect_profiles.sh <tag> [bench args...]
set -u
tag=$1; shift
raw=gpurun_out/profiles_raw/$tag
mkdir -p $raw
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { # name, rocprof args...
  name=$1; shift
  mkdir -p $raw/$name
  timeout 600 rocprofv3 "$@" --output-format csv -d $raw/$name -- python3 bench.py --no-cpu "${BENCH_ARGS[@]}" > $raw/$name.log 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS=("$@" --steps 100 --warmup 20)
run stats --kernel-trace --stats
BENCH_ARGS=("$@" --steps 20 --warmup 5 --no-events)
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
tail -1 $raw/stats.log | cut -c1-400
