#!/bin/bash
# kernel trace of the 50/50 workload (BASELINE configs[3]) -- tools/r06_prof_zipf50.sh [steps] [extra bench flags]
cd /tmp && export TMPDIR=/tmp
STEPS=${1:-20}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_zipf50_$STEPS
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o z50 -- python3 $GRAFT_REPO_ROOT/bench.py --workload zipf50 --steps $STEPS --warmup 5 --no-live-traffic --no-cpu --no-pcie --no-abi --repeats 1 "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)
for fn in f:
    print(fn)
    rows=list(csv.DictReader(open(fn)))
    for r in rows[:16]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} pct {r['Percentage']}")
PY
tail -2 $OUT/bench.log | cut -c1-400
