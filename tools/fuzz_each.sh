#!/bin/bash
# every fuzz case in a process of its own under a short time limit: a kernel that never ends costs
# 40 s, not the session.   usage: fuzz_each.sh <test name> <first> <last> [env ...]
name=$1; lo=$2; hi=$3; shift 3
bad=0
for i in $(seq $lo $hi); do
  env "$@" timeout 40 python -m pytest "tests/test_gpu_fuzz.py::${name}[$i]" -x -q > /tmp/fz.txt 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "case $i: rc $rc $(grep -E 'passed|failed|Error' /tmp/fz.txt | tail -1)"; bad=$((bad+1)); [ $rc -eq 124 ] && break; fi
done
echo "$name $lo..$hi: $bad bad"
