#!/bin/bash
cd "$GRAFT_REPO_ROOT"; d=gpurun_out/s38; mkdir -p $d
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $d/pytest.txt
for w in 1 2 4 8; do for x in 0 7 15; do echo "== zipf loopback world $w fake_xfer_us $x (one group launch of 32 work-groups)"; MKV_SHARD_FAKE_XFER_US=$x timeout 300 python tools/loopback_trace.py $w 64 30 zipf 2>&1 | grep "us per batch" | tail -1; done; done | tee $d/loopback_pred.txt
for g in 8 16 64 0; do echo "== world 8, fake 0, MKV_SHARD_LOOP_WGS=$g"; MKV_SHARD_LOOP_WGS=$g timeout 300 python tools/loopback_trace.py 8 64 30 zipf 2>&1 | grep "us per batch" | tail -1; done | tee -a $d/loopback_pred.txt
MKV_SHARD_TRACE=6 timeout 300 python tools/loopback_trace.py 8 64 30 zipf 2> $d/loop8_trace.txt | tail -1
for a in "zipf95 new" "zipf95 upd" "zipf50 new" "zipf50 upd" "zipf50 new 6 0.5"; do echo "== tail phases $a"; timeout 300 python tools/tail_phases.py $a 2>&1 | grep "^step" ; done | tee $d/tail_phases.txt
{
echo "== box: $(hostname) $(date -u +%FT%TZ)"
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('driver command run $i: value', j['value'], 'repeats', j['repeats']['mops'], 'spread', j['repeats']['spread'], 'abi_three_call', j['abi_three_call']['value'], 'launch_us', j['roofline']['launch_us'])"; done
for i in 1 2 3 4 5; do python bench.py --no-cpu --no-pcie 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('default command (--no-cpu --no-pcie) run $i: value', j['value'], 'repeats', j['repeats']['mops'], 'spread', j['repeats']['spread'], 'abi_three_call', j['abi_three_call']['value'], 'launch_us', j['roofline']['launch_us'])"; done
} 2>&1 | tee $d/repeat.txt
