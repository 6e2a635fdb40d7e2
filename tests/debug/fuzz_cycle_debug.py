#!/usr/bin/env python3
"""Replay one seed of tests/test_gpu_fuzz.py::test_fuzz_cycles and, at the first table mismatch,
print the insert requests that touch the first differing bucket.  usage: fuzz_cycle_debug.py <seed> [tuning=value ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from megakv_amd import Geometry, HashIndex  # noqa: E402
from megakv_amd.lib import load_library  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.gpu_util import dev, host, ogeom, random_table, random_requests, split_blocks  # noqa: E402

seed = int(sys.argv[1])
over = dict(a.split("=") for a in sys.argv[2:])
rng = np.random.default_rng(9000 + seed)
L = load_library()
mem_p = int(rng.integers(10, 19))
iblock_p = int(rng.integers(0, min(5, mem_p - 7) + 1))
scheme = int(rng.integers(0, 2))
g = Geometry(mem_p=mem_p, scheme=scheme, iblock_p=iblock_p, max_cuckoo=int(rng.choice([1, 2, 5])))
og = ogeom(g)
L.mkv_set_tuning(b"tail_rounds", int(rng.choice([0, 3, 48])))
L.mkv_set_tuning(b"insert_chunk", int(rng.choice([1024, 8192, 65536])))
L.mkv_set_tuning(b"tail_groups", int(rng.choice([1, 5, 32])))
L.mkv_set_tuning(b"grid_min", int(rng.choice([0, 64, 2048])))      # (diagnostics build only)
for k, v in over.items():
    L.mkv_set_tuning(k.encode(), int(v))
sig_space = int(rng.choice([40, 5000, 1 << 20]))
exp = random_table(g, rng, float(rng.choice([0.0, 0.3, 0.7, 0.9])), sig_space)
idx = HashIndex(g)
idx.load_table(exp)
nslots = g.nbuckets * 8
memset = bool(rng.integers(0, 2))
overlap = str(rng.choice(["fused", "fused", "side", "none"]))
print(dict(mem_p=mem_p, iblock_p=iblock_p, scheme=scheme, max_cuckoo=g.max_cuckoo, sig_space=sig_space, overlap=overlap))
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for step in range(3):
        nq = int(rng.integers(0, 20000))
        nd = int(rng.integers(0, 3000))
        ni = int(rng.integers(0, max(2, min(30000, nslots))))
        hot = float(rng.choice([0.0, 0.1, 0.5]))
        q = random_requests(g, rng, nq, sig_space, hot=hot, zeros=float(rng.choice([0.0, 0.03])))[:, :2].copy()
        d = random_requests(g, rng, nd, sig_space)
        e = random_requests(g, rng, ni, sig_space, hot=hot, zero_low=float(rng.choice([0.0, 0.1])),
                            zeros=float(rng.choice([0.0, 0.03])))
        occ = np.argwhere(exp[:, :8] != 0)
        if len(occ):
            pick = occ[rng.integers(0, len(occ), size=min(nq, nd) // 2)]
            q[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
            q[: len(pick), 1] = pick[:, 0]
            d[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
            d[: len(pick), 1] = pick[:, 0]
            d[: len(pick), 2] = exp[pick[:, 0], 8 + pick[:, 1]]
        how = str(rng.choice(["single", "region", "uneven"]))
        blocks = split_blocks(e, g, rng, how)
        dblocks = [dev(b.reshape(-1, 3)) for b in blocks]
        ptrs, cnts, keep = idx.stage_blocks(dblocks)
        out = dev(np.full((max(nq, 1), 2), 0x5555AAAA, np.uint32))
        dq, dd = (dev(q) if nq else None), (dev(d) if nd else None)
        hint = int(rng.choice([0, ni]))
        idx.cycle(dq, out if nq else None, ptrs, cnts, len(keep), delem=dd, memset=memset, overlap=overlap,
                  stream=stream, insert_hint=hint)
        before = exp.copy()
        if nd:
            O.delete(og, exp, d)
        after_delete = exp.copy()
        O.insert(og, exp, blocks)
        stream.synchronize()
        got = idx.table_numpy()
        try:
            info = idx.last_update_info(stream, cycle_insert=(overlap != "none" and ni > 0))
        except Exception as ex:                       # (no insert half ran through that workspace yet)
            info = str(ex)
        print("step", step, dict(nq=nq, nd=nd, ni=ni, how=how, hint=hint), info)
        if not np.array_equal(got, exp):
            bad = np.argwhere((got != exp).any(axis=1))[:, 0]
            print("differing buckets", bad[:10], "of", len(bad))
            allreq = np.concatenate([b.reshape(-1, 3) for b in blocks if len(b)])
            hm, bm = g.hash_mask, g.block_hash_mask
            P = allreq[:, 1] & hm
            A = (((allreq[:, 1] ^ allreq[:, 0]) & bm) | (allreq[:, 1] & ~np.uint32(bm))) & hm
            for b in bad[:2]:
                print("bucket", b, "\n before ", before[b], "\n deleted", after_delete[b], "\n oracle ", exp[b], "\n gpu    ", got[b])
                sel = np.flatnonzero((P == b) | (A == b))
                print("   requests on it:", len(sel), "first", sel[:5], "last", sel[-5:])
                for sl in range(8):
                    if got[b][8 + sl] != exp[b][8 + sl]:
                        print("   slot", sl, "oracle loc from idx", np.flatnonzero(allreq[:, 2] == exp[b][8 + sl]), "gpu loc from idx", np.flatnonzero(allreq[:, 2] == got[b][8 + sl]))
                bounds = np.cumsum([0] + [len(x) for x in blocks])
                print("   block bounds", bounds.tolist())
                for i in sel[-12:]:
                    freeP = int((after_delete[P[i], :8] == 0).sum())
                    freeP0 = int((before[P[i], :8] == 0).sum())
                    print(f"   idx {i} sig {allreq[i,0]} P {P[i]} A {A[i]} loc {allreq[i,2]} free(P) before {freeP0} after-delete {freeP}")
            break
