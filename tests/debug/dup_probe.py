#!/usr/bin/env python3
"""One hot key many times in one batch (a few other keys on the same bucket in between), through
mkv_insert and mkv_cycle, one block or sixteen: which location ends up in the table?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from megakv_amd import Geometry, HashIndex
from oracle import oracle as O
from tests.gpu_util import dev, ogeom, split_blocks

for scheme in (0, 1):
    for nblk_mode in ("single", "region"):
        for path in ("insert", "cycle", "cycle+delete"):
            for lead in (0, 491):
                g = Geometry(mem_p=14, scheme=scheme, iblock_p=4)
                og = ogeom(g)
                rng = np.random.default_rng(5)
                n = 137
                e = np.zeros((lead + n, 3), np.uint32)
                if lead:
                    e[:lead, 0] = rng.integers(1000, 1 << 30, size=lead)
                    e[:lead, 1] = rng.integers(0, 1 << 32, size=lead, dtype=np.uint64)
                    e[:lead, 2] = rng.integers(1, 1 << 32, size=lead, dtype=np.uint64)
                e[lead:, 0] = 18
                e[lead:, 1] = 138
                e[lead:, 2] = rng.integers(1, 1 << 32, size=n, dtype=np.uint64)
                e[lead + 12, 0] = 24
                e[lead + 40, 0] = 10
                blocks = split_blocks(e, g, rng, nblk_mode)
                exp = O.new_table(og)
                idx = HashIndex(g)
                d = np.zeros((50, 3), np.uint32); d[:, 0] = 7; d[:, 1] = rng.integers(0, 1 << 32, size=50, dtype=np.uint64); d[:, 2] = 9
                O.insert(og, exp, blocks)
                dblocks = [dev(b.reshape(-1, 3)) for b in blocks]
                if path == "insert":
                    idx.insert(dblocks)
                else:
                    ptrs, cnts, keep = idx.stage_blocks(dblocks)
                    q = dev(np.ascontiguousarray(e[:64, :2]))
                    out = torch.zeros((64, 2), dtype=torch.int32, device="cuda")
                    idx.cycle(q, out, ptrs, cnts, len(keep), delem=dev(d) if path == "cycle+delete" else None, insert_hint=len(e))
                torch.cuda.synchronize()
                got = idx.table_numpy()
                ok = np.array_equal(got, exp)
                bad = np.argwhere((got != exp).any(axis=1))[:, 0]
                msg = ""
                if not ok:
                    b = bad[0]
                    all_ = np.concatenate([x.reshape(-1, 3) for x in blocks if len(x)])
                    sl = np.flatnonzero(got[b][8:] != exp[b][8:])
                    msg = f"bucket {b} slots {sl.tolist()} gpu loc from idx {[np.flatnonzero(all_[:, 2] == got[b][8 + s]).tolist() for s in sl]} oracle from {[np.flatnonzero(all_[:, 2] == exp[b][8 + s]).tolist() for s in sl]}"
                print(f"scheme {scheme} {nblk_mode:7s} {path:13s} lead {lead:4d}: {'ok' if ok else 'DIFF ' + msg}")
