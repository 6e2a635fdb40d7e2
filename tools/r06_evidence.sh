#!/bin/bash
# Round 6's committed evidence from the final library, on one box: the profiler passes behind profiles/r06_* and the results table
set -u
HEAD_ID=${1:-unknown}
SPECS=("driver:zipf95-mix:5:20:--steps 20 --warmup 5" "default:zipf95-mix:20:100:" "get100:get100-mix:20:100:--workload get100"
       "zipf50:zipf50-mix:5:20:--workload zipf50 --steps 20 --warmup 5" "supd:zipf95-upd:5:20:--sets upd --steps 20 --warmup 5"
       "sharded1:zipf95-mix-sharded:20:100:--force-sharded")
for spec in "${SPECS[@]}"; do
  case " ${ONLY:-} " in "  ") ;; *" ${spec%%:*} "*) ;; *) continue;; esac      # ONLY="zipf50 supd": those tags only
  IFS=: read tag key W K args <<< "$spec"
  echo "=== $tag ($args)"
  PMC=${PMC:-1} bash tools/collect_profiles.sh $tag $args
  python3 tools/make_profiles.py r06 $tag $key $W $K $HEAD_ID 3 $args > gpurun_out/profiles_raw/$tag.summary 2>&1 || echo "make_profiles $tag failed"
done
# what comes back from the box is gpurun_out/ and at most 64 MiB of it: the summaries, not the raw traces
mkdir -p gpurun_out/profiles_r06 && cp profiles/r06_*_kernel_stats.csv profiles/r06_*_timed_region.json profiles/search_traffic.json gpurun_out/profiles_r06/ 2>/dev/null
cp gpurun_out/profiles_raw/*.summary gpurun_out/profiles_r06/ 2>/dev/null
for t in gpurun_out/profiles_raw/*/; do n=$(basename $t); tail -1 $t/stats.json > gpurun_out/profiles_r06/$n.line.json 2>/dev/null; tail -3 $t/stats.err > gpurun_out/profiles_r06/$n.err 2>/dev/null; done
rm -rf gpurun_out/profiles_raw
[ -z "${NO_TABLE:-}" ] && bash tools/results_table.sh
