#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s32; mkdir -p $d
p() { python - "$1" "$2" <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], x["value"], x["ms_per_step"], x.get("roofline",{}).get("launch_us"))
except Exception as e:
    print(sys.argv[2], "bad", e)
PY
}
for mb in 128 205 256 384; do
for cfg in "lf085 --steps 40 --warmup 5 --lf 0.85" "drv --steps 20 --warmup 5" "lf05 --steps 20 --warmup 5 --lf 0.5"; do
  set -- $cfg; name=$1; shift
  MKV_MARK_BLOCKS=$mb timeout 300 python bench.py --no-cpu --no-pcie --no-abi "$@" > $d/mb${mb}_$name.json 2>> $d/r4.err; p $d/mb${mb}_$name.json mb${mb}_$name
done; done
