#!/bin/bash
# (round 6) what the tail launch executes: SQ instruction counters per k_tail dispatch of a bench command (one PMC pass each)
#   tools/r06_tail_insts.sh <tag> [bench args]   -> gpurun_out/tail_insts_<tag>.txt
set -u
tag=$1; shift
raw=gpurun_out/tail_insts_raw_$tag
rm -rf $raw; mkdir -p $raw
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  d=$raw/$(echo $c | tr ' ' '_'); mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --no-cpu --no-pcie --no-live-traffic --no-abi --no-sched "$@" > $d.json 2> $d.err
  echo "$c rc=$?"
done
python3 - "$raw" > gpurun_out/tail_insts_$tag.txt <<'PY'
import csv,glob,sys,collections
raw=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(raw+'/**/*counter_collection.csv',recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        name='k_tail' if 'k_tail' in k else 'k_round' if 'k_round' in k else 'k_search_mark' if 'k_search_mark' in k else None
        if not name: continue
        per[(name,r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
    for (name,d,c),v in per.items():
        acc[name][c].append(v)
for name in acc:
    for c,vals in sorted(acc[name].items()):
        vals=vals[-20:]
        print(f"{name:14s} {c:20s} per dispatch (last {len(vals)}): mean {sum(vals)/len(vals):14.0f}  max {max(vals):14.0f}")
PY
rm -rf $raw
cat gpurun_out/tail_insts_$tag.txt
