# usage: z50_sweep.sh  -- 50/50 cuckoo bench over chunk sizes / calm factors, 20 and 60 steps
for cfg in "65536 1" "65536 2" "65536 4" "32768 4" "131072 1"; do
  set -- $cfg
  for st in 20 60; do
    echo -n "chunk=$1 calm=$2 steps=$st: "; MKV_INSERT_CHUNK=$1 MKV_CALM_FACTOR=$2 python3 bench.py --workload zipf50 --steps $st --warmup 5 --no-cpu --no-pcie 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
