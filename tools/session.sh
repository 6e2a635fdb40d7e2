#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for sp in 0 1; do for w in 1 8; do echo "== spin $sp world $w"; MKV_SHARD_SPIN=$sp python3 tools/loopback_trace.py $w 64 30 zipf 2>/dev/null | tail -2; done; done
for sp in 0 1; do MKV_SHARD_SPIN=$sp python3 bench.py --force-sharded --no-cpu --no-pcie 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spin $sp sharded1', d['value'], round(d['ms_per_step']*1e3,2), d['phases_us_per_rank'])"; done
