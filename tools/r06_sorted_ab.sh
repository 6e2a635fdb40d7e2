python -m pytest tests/test_gpu_sorted.py -q -x 2>&1 | tail -5
for sm in 0 131072; do
  for st in 20 100; do
    echo "== zipf50 sorted_min=$sm steps=$st"
    MKV_SORTED_MIN=$sm python bench.py --workload zipf50 --steps $st --warmup 5 --no-live-traffic --no-cpu --no-pcie --no-abi 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('runs'))
"
  done
done
echo "== default"
python bench.py --steps 20 --warmup 5 --no-live-traffic --no-cpu --no-pcie --no-abi 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
"
