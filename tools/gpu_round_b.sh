#!/bin/bash
# GPU round B: parity of the reworked insert path, then its rates
set -u
out=gpurun_out/r2b; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench rc=$?"
timeout 300 python tools/insert_probe.py 0.5 524288 4 > $out/probe_lf50.log 2>&1
timeout 300 python tools/insert_probe.py 0.85 52428 4 > $out/probe_lf85.log 2>&1
timeout 600 python bench.py --workload zipf50 --steps 100 --warmup 5 --no-cpu --no-pcie > $out/bench_z50_100.json 2> $out/bench_z50_100.err
timeout 600 python bench.py --lf 0.5 --no-cpu --no-pcie > $out/bench_lf50.json 2> $out/bench_lf50.err
timeout 600 python bench.py --lf 0.85 --steps 20 --no-cpu --no-pcie > $out/bench_lf85.json 2> $out/bench_lf85.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('launch_us'), r.get('frac'), d.get('get_loc_correct_last_timed'), d.get('set_loc_correct_last_timed'), d.get('table_entries'))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -4 $out/probe_lf50.log | cut -c1-400; tail -4 $out/probe_lf85.log | cut -c1-400
