#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s21; mkdir -p $d
for cfg in "MKV_SHARD_SVC=2 MKV_SHARD_LANES=3" "MKV_SHARD_SVC=3 MKV_SHARD_LANES=3" "MKV_SHARD_SVC=3 MKV_SHARD_LANES=4" "MKV_SHARD_SVC=2 MKV_SHARD_LANES=4" "MKV_SHARD_SVC=1 MKV_SHARD_LANES=3" "MKV_SHARD_SVC=3 MKV_SHARD_LANES=3 MKV_SHARD_PACED=0"; do
  for w in 8 2; do
    echo "== $cfg world $w fake 7"; env $cfg MKV_SHARD_FAKE_XFER_US=7 timeout 300 python tools/loopback_trace.py $w 64 30 zipf 2>&1 | grep "us per batch\|rror" | tail -1
  done
done 2>&1 | tee $d/svc.txt
for cfg in "MKV_SHARD_SVC=1 MKV_SHARD_LANES=2" "MKV_SHARD_SVC=2 MKV_SHARD_LANES=2" "MKV_SHARD_SVC=2 MKV_SHARD_LANES=3"; do
  echo "== $cfg world 1"; env $cfg timeout 300 python tools/loopback_trace.py 1 64 30 zipf 2>&1 | grep "us per batch\|rror" | tail -1
done 2>&1 | tee -a $d/svc.txt
