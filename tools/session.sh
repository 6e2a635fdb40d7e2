#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh driver --steps 20 --warmup 5 2>&1 | tail -6
bash tools/collect_profiles.sh default 2>&1 | tail -6
PMC=1 bash tools/collect_profiles.sh get100 --workload get100 --no-pcie 2>&1 | tail -6
PMC=1 bash tools/collect_profiles.sh zipf50 --workload zipf50 --steps 20 --warmup 5 --no-cpu --no-pcie 2>&1 | tail -6
du -sh gpurun_out/profiles_raw
# keep what make_profiles.py needs, drop the rest (the merge back is limited to 64 MiB)
find gpurun_out/profiles_raw -name "*agent_info*" -delete
find gpurun_out/profiles_raw -name "*domain_stats*" -delete
du -sh gpurun_out/profiles_raw
