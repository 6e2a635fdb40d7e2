#!/bin/bash
# Runs on the GPU box: the seeded fuzz test over many seeds and forced tail shapes.
# usage: fuzz_campaign.sh [cases]
n=${1:-2000}
export MKV_FUZZ_CASES=$n
run() { echo -n "$*: "; env "$@" timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -1; }
run MKV_X=0
run MKV_FUZZ_TAIL_GROUPS=1 MKV_FUZZ_TAIL_ROUNDS=48 MKV_FUZZ_GRID_MIN=1000000000
run MKV_FUZZ_TAIL_GROUPS=32 MKV_FUZZ_TAIL_ROUNDS=48 MKV_FUZZ_GRID_MIN=0
run MKV_FUZZ_TAIL_GROUPS=7 MKV_FUZZ_TAIL_ROUNDS=48 MKV_FUZZ_GRID_MIN=16 MKV_FUZZ_INSERT_CHUNK=1024
run MKV_FUZZ_TAIL_GROUPS=32 MKV_FUZZ_TAIL_ROUNDS=3 MKV_FUZZ_INSERT_CHUNK=2048 MKV_FUZZ_CALM_FACTOR=1
run MKV_FUZZ_TAIL_GROUPS=2 MKV_FUZZ_TAIL_ROUNDS=0 MKV_FUZZ_INSERT_CHUNK=65536
# how order violations of late chains are taken back: always the general way / the region suffix
run MKV_FUZZ_DEBUG_SKIP=1 MKV_FUZZ_TAIL_GROUPS=32 MKV_FUZZ_TAIL_ROUNDS=48
run MKV_FUZZ_DEBUG_SKIP=2 MKV_FUZZ_TAIL_GROUPS=5 MKV_FUZZ_TAIL_ROUNDS=48 MKV_FUZZ_INSERT_CHUNK=8192
run MKV_FUZZ_DEBUG_SKIP=2 MKV_FUZZ_TAIL_GROUPS=2 MKV_FUZZ_GRID_MIN=0 MKV_FUZZ_CALM_FACTOR=1 MKV_FUZZ_ROUND_BLOCKS=1
# rounds 1 and 2 on few work-groups (their queues of held-back requests run over), every chain left to the tail
run MKV_FUZZ_ROUND_BLOCKS=2 MKV_FUZZ_DEBUG_SKIP=4 MKV_FUZZ_TAIL_GROUPS=32
run MKV_FUZZ_ROUND_BLOCKS=1 MKV_FUZZ_DEBUG_SKIP=0
# round 4: round 2 always inside the tail launch / never; no self-help of second registrants
run MKV_FUZZ_MERGE_ROUND2=2 MKV_FUZZ_SELF_HELP=1 MKV_FUZZ_TAIL_GROUPS=32
run MKV_FUZZ_MERGE_ROUND2=0 MKV_FUZZ_SELF_HELP=0 MKV_FUZZ_TAIL_GROUPS=5
# round 5: round 1 never / always carries out the mark pass's plans; cycles (the mark pass inside the search launch, deletes between it and round 1)
run MKV_FUZZ_PLANS=0 MKV_FUZZ_TAIL_GROUPS=32
run MKV_FUZZ_PLANS=1 MKV_FUZZ_ROUND_BLOCKS=512 MKV_FUZZ_MERGE_ROUND2=1
# round 6: big insert passes forced at every size (bins, apply per bucket, round 2 over the left-overs, the tail's fourth source)
run MKV_FUZZ_SORTED_MIN=1 MKV_FUZZ_INSERT_CHUNK=65536 MKV_FUZZ_TAIL_GROUPS=32
run MKV_FUZZ_SORTED_MIN=1 MKV_FUZZ_INSERT_CHUNK=65536 MKV_FUZZ_DEBUG_SKIP=2 MKV_FUZZ_TAIL_GROUPS=2 MKV_FUZZ_ROUND_BLOCKS=1
run MKV_FUZZ_SORTED_MIN=1 MKV_FUZZ_INSERT_CHUNK=65536 MKV_FUZZ_DEBUG_SKIP=1 MKV_FUZZ_TAIL_ROUNDS=0 MKV_FUZZ_BIG=1
