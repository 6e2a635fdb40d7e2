#!/bin/bash
# One GPU session (gpurun -- bash tools/session.sh): rewritten for every call; what it printed is kept
# under gpurun_out/.
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s3; mkdir -p $d
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --update-info > $d/$name.json 2> $d/$name.err; \
	python - $d/$name.json $name <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], x["value"], x["ms_per_step"], x.get("roofline", {}).get("launch_us"), x.get("set_loc_correct_last_timed"), x.get("table_entries_ok"))
except Exception as e:
    print(sys.argv[2], "bad", e)
PY
}
# search occupancy: blocks per CU limited by dynamic LDS (12.3 KB static): 160 KB / (12.3 + x)
run occ_base MKV_PLAN=0
run occ_6 MKV_PLAN=0 MKV_DBG_DYNLDS=14000
run occ_5 MKV_PLAN=0 MKV_DBG_DYNLDS=20000
run occ_4 MKV_PLAN=0 MKV_DBG_DYNLDS=28000
run occ_3 MKV_PLAN=0 MKV_DBG_DYNLDS=41000
run occ_2 MKV_PLAN=0 MKV_DBG_DYNLDS=52000
run occ_base2 MKV_PLAN=0
run u2_occ4 MKV_PLAN=0 MKV_DBG_DYNLDS=28000 MKV_MARK_BLOCKS=256
run mb256 MKV_PLAN=0 MKV_MARK_BLOCKS=256
