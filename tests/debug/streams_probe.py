#!/usr/bin/env python3
"""tests/test_gpu_abi.py::test_updates_of_one_table_from_several_streams..., with knobs:
usage: streams_probe.py <nstreams> [tuning=value ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import megakv_amd as M
from megakv_amd.lib import load_library
from oracle import oracle as O
ns = int(sys.argv[1])
L = load_library(diagnostics=True)      # (the diagnostics build: make -C megakv_amd/csrc DIAG=1)
for a in sys.argv[2:]:
    k, v = a.split("="); L.mkv_set_tuning(k.encode(), int(v))
g = M.Geometry(mem_p=20); og = O.geom(mem_p=20)
idx = M.HashIndex(g, lib=L); exp = O.new_table(og)
rng = np.random.default_rng(12)
streams = [torch.cuda.Stream() for _ in range(ns)]
pool = rng.integers(1, 2**32, size=(30000, 3), dtype=np.uint64).astype(np.uint32)
keep = []
for call in range(12):
    e = pool[rng.integers(0, len(pool), size=20000)].copy()
    e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)
    s = streams[call % ns]
    before = exp.copy()
    if call % 4 == 3:
        occ = np.argwhere(exp[:, :8] != 0)[::7]
        d = np.stack([exp[occ[:, 0], occ[:, 1]], occ[:, 0].astype(np.uint32), exp[occ[:, 0], 8 + occ[:, 1]]], axis=1)
        d = np.ascontiguousarray(d.astype(np.uint32)); O.delete(og, exp, d)
        t = torch.from_numpy(d.view(np.int32)).cuda(); keep.append(t); idx.delete(t, stream=s)
    else:
        O.insert(og, exp, [e])
        t = torch.from_numpy(e.view(np.int32)).cuda(); keep.append(t)
        ptrs, cnts, kk = idx.stage_blocks([t]); keep.append((ptrs, cnts, kk))
        idx.insert_staged(ptrs, cnts, 1, stream=s, hint=len(e))
    if "sync" in os.environ.get("PROBE", ""):
        torch.cuda.synchronize()
        got = idx.table_numpy()
        if not np.array_equal(got, exp):
            bad = np.argwhere((got != exp).any(axis=1))[:, 0]
            print("call", call, "differs in", len(bad), "buckets", bad[:5], idx.last_update_info(s)); b = bad[0]
            print(" before", before[b], "\n oracle", exp[b], "\n gpu   ", got[b])
            hm, bm = g.hash_mask, g.block_hash_mask
            P = e[:, 1] & hm
            A = (((e[:, 1] ^ e[:, 0]) & bm) | (e[:, 1] & ~np.uint32(bm))) & hm
            import ctypes
            for i in np.flatnonzero((P == b) | (A == b)):
                o = (ctypes.c_uint32 * 9)()
                L.mkv_debug_request(0, int(i), o)
                print(f"   idx {i}: tag {o[0]} flags {o[4] & 0xff:#x} grp {o[4] >> 8} undo {[hex(x) for x in o[5:9]]}")
            for i in np.flatnonzero((P == b) | (A == b)):
                print(f"   idx {i} sig {e[i,0]} hash {e[i,1]:#x} P {P[i]} A {A[i]} loc {e[i,2]} | P row before {before[P[i]][:8]} | A row before {before[A[i]][:8]}")
            for sl in range(8):
                if got[b][8 + sl] != exp[b][8 + sl]:
                    print("   slot", sl, "oracle loc from idx", np.flatnonzero(e[:, 2] == exp[b][8 + sl]), "gpu loc from idx", np.flatnonzero(e[:, 2] == got[b][8 + sl]))
            break
torch.cuda.synchronize()
got = idx.table_numpy()
bad = np.argwhere((got != exp).any(axis=1))[:, 0]
print("streams", ns, sys.argv[2:], "differing buckets", len(bad))
