#!/bin/bash
# CPU-side undefined-behaviour check (GPU sanitizers are not available on the pool): the oracle and the
# workload generator are rebuilt with gcc -fsanitize=undefined in place of their shared objects, the CPU
# tests that exercise them are run, and the normal builds are put back.  usage: tools/ubsan_cpu.sh
set -u
cd "$(dirname "$0")/.."
O=oracle/_build/libgpuhash_oracle.so
W=workload/_lib/libmkv_workload.so
cp $O /tmp/_oracle_keep.so && cp $W /tmp/_workload_keep.so || exit 1
restore() { cp /tmp/_oracle_keep.so $O; cp /tmp/_workload_keep.so $W; }
trap restore EXIT
F="-O1 -g -fsanitize=undefined -fno-sanitize-recover=undefined -fPIC -shared"
gcc $F -pthread -o $O oracle/gpuhash_oracle.c || exit 1
gcc $F -o $W workload/workload.c -lm || exit 1
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_oracle_kat.py tests/test_packets.py \
	tests/test_workload.py -q -x 2>&1 | grep -E " passed| failed|runtime error" | tail -5
