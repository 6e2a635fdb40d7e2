#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== under torch.distributed.run, one rank, the sharded engine over RCCL (transport check in a child first)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-sharded --steps 30 --warmup 5 --no-cpu --no-pcie --mem-p 24 --batch-log2 16 2>gpurun_out/tr_err.txt | tail -1 | cut -c1-300
grep -i "preflight\|transport\|error" gpurun_out/tr_err.txt | head -5
bash tools/collect_profiles.sh driver --steps 20 --warmup 5 2>&1 | tail -2 | cut -c1-200
bash tools/collect_profiles.sh default 2>&1 | tail -1 | cut -c1-200
bash tools/collect_profiles.sh get100 --workload get100 --no-pcie 2>&1 | tail -1 | cut -c1-200
bash tools/collect_profiles.sh zipf50 --workload zipf50 --steps 20 --warmup 5 --no-cpu --no-pcie 2>&1 | tail -1 | cut -c1-200
find gpurun_out/profiles_raw -name "*agent_info*" -delete; find gpurun_out/profiles_raw -name "*domain_stats*" -delete
for t in lf05 lf085 sharded1 supd z50lf04 z50lf05 zipf50upd zipf95_mix get100_mix; do rm -rf gpurun_out/profiles_raw/$t; done
du -sh gpurun_out/profiles_raw
