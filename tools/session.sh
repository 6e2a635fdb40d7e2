#!/bin/bash
# tail phases, S-upd: tree's diagnostics build vs r05's
lib=megakv_amd/_lib/libmegakv_hip_diag.so
cp $lib /tmp/libd_tree.so
for v in tree r05 tree r05; do
  [ $v = tree ] && cp /tmp/libd_tree.so $lib || cp _ab/libmegakv_hip_r05diag.so $lib
  old=""; [ $v = r05 ] && old=1
  echo "== $v"
  MKV_LIB_OLD_BUILD=$old python3 tools/tail_phases.py zipf95 upd 8 2>&1 | tail -6 | sed 's/helped.*tail_parallel/tail_parallel/' | cut -c1-330
done
cp /tmp/libd_tree.so $lib
