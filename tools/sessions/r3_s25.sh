#!/bin/bash
# final tree: the default and the driver's command profiled again (kernel stats + PMC), whole GPU suite
bash tools/collect_profiles.sh default
bash tools/collect_profiles.sh driver --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh zipf50upd --workload zipf50 --sets upd --steps 20 --warmup 5
PMC=0 bash tools/collect_profiles.sh sharded1 --force-sharded --steps 20 --warmup 5
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-pcie 2>/dev/null | cut -c1-200; done
