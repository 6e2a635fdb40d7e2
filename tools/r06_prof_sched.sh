#!/bin/bash
# kernel trace of the plain-C scheduler host (examples/scheduler_cycle.c), deferred mode:  tools/r06_prof_sched.sh [mapped 0|1] [onecall 0|1]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sched_${1:-0}_${2:-0}
rm -rf $OUT; mkdir -p $OUT
export MKV_DEFER=${MKV_DEFER:-1}
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o sc -- $GRAFT_REPO_ROOT/megakv_amd/_lib/scheduler_cycle 100 16 ${1:-0} ${2:-0} 0 > $OUT/run.log 2>&1
grep -E "cycles|scheduler section" $OUT/run.log | cut -c1-200
python3 - <<PY
import csv,glob
for fn in glob.glob("$OUT/*kernel_stats.csv")+glob.glob("$OUT/*memory_copy_stats.csv"):
    rows=list(csv.DictReader(open(fn)))
    for r in rows[:12]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
