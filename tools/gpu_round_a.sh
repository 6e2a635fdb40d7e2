#!/bin/bash
# GPU round A: full test suite, the driver's bench command, probes of today's insert path
set -u
out=gpurun_out/r2a; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest.log | cut -c1-300
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err; echo "bench driver $i rc=$?" | tee -a $out/summary.txt
done
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.txt
timeout 600 python bench.py --workload get100 --no-cpu --no-pcie > $out/bench_get100.json 2> $out/bench_get100.err
timeout 300 python tools/insert_probe.py 0.5 524288 4 > $out/probe_lf50.log 2>&1
timeout 300 python tools/insert_probe.py 0.85 52428 4 > $out/probe_lf85.log 2>&1
timeout 600 python bench.py --workload zipf50 --steps 60 --warmup 5 --no-cpu --no-pcie > $out/bench_z50_60.json 2> $out/bench_z50_60.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('launch_us'), r.get('avg_launch_us'), r.get('p10_us'), r.get('p90_us'), r.get('frac'), d.get('set_loc_correct_last_timed'), d.get('table_entries'), (d.get('step_algorithmic') or {}).get('GBps'))
    except Exception as e:
        print(f, 'ERR', e)
PY
