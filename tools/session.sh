#!/bin/bash
# insert-only A/B: tree's diagnostics build vs r05's (tools/insert_probe.py)
lib=megakv_amd/_lib/libmegakv_hip_diag.so
cp $lib /tmp/libd_tree.so
for v in tree r05 tree r05; do
  [ $v = tree ] && cp /tmp/libd_tree.so $lib || cp _ab/libmegakv_hip_r05diag.so $lib
  old=""; [ $v = r05 ] && old=1
  echo "== $v"
  MKV_LIB_OLD_BUILD=$old python3 tools/insert_probe.py 0.2 524288 5 new 2>&1 | grep "^batch" | cut -c1-22
done
cp /tmp/libd_tree.so $lib
