#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace/stats of the bench command
# plus separate PMC passes (FETCH_SIZE / WRITE_SIZE / L2 counters, each in its own run,
# --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Raw output ->
# gpurun_out/profiles_raw/<tag>/ ; tools/make_profiles.py turns it into the committed
# summaries under profiles/.
# usage: tools/collect_profiles.sh <tag> [bench args...]      (tag "default": no bench args = the default command)
set -u
tag=$1; shift
raw=gpurun_out/profiles_raw/$tag
rm -rf $raw; mkdir -p $raw
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { # name, rocprof args...
  name=$1; shift
  mkdir -p $raw/$name
  timeout 900 rocprofv3 "$@" --output-format csv -d $raw/$name -- python3 bench.py "${BENCH_ARGS[@]}" > $raw/$name.json 2> $raw/$name.err
  echo "$name rc=$?"
}
# the stats run is the EXACT command (side legs included); the counter passes leave the side
# legs out (their searches of other buffers would mix into the per-kernel averages)
BENCH_ARGS=("$@")
run stats --kernel-trace --stats
BENCH_ARGS=("$@" --no-cpu --no-pcie)
if [ "${PMC:-1}" = "0" ]; then tail -1 $raw/stats.json | cut -c1-300; exit 0; fi   # PMC=0: the stats run only
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
tail -1 $raw/stats.json | cut -c1-300
