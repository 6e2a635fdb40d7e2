#!/bin/bash
# A/B on one box: the round-2 tree (copied to _ab/r2tree, library built from commit 2259bb3) against this tree
set -u
cd "$GRAFT_REPO_ROOT"
one() { # dir args...
  d=$1; shift
  (cd $d && timeout 600 python3 bench.py --no-cpu --no-pcie "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$d', d['value'], round(d['ms_per_step']*1e3,2), r.get('launch_us'))")
}
for rep in 1 2 3; do
  one _ab/r2tree "$@"
  one . "$@"
done
