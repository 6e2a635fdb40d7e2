#!/usr/bin/env python3
"""Replay one seed of tests/test_gpu_fuzz.py and, at the first table mismatch, print the
requests that touch the first differing bucket (canonical index, sig, P, A, loc).
usage: fuzz_debug.py <seed> [tuning=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from megakv_amd import Geometry, HashIndex  # noqa: E402
from megakv_amd.lib import load_library  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.gpu_util import dev, ogeom, random_table, random_requests, split_blocks  # noqa: E402


def main():
    seed = int(sys.argv[1])
    over = dict(a.split("=") for a in sys.argv[2:])
    rng = np.random.default_rng(1000 + seed)
    L = load_library(diagnostics=True)
    mem_p = int(rng.integers(10, 19))
    iblock_p = int(rng.integers(0, min(5, mem_p - 7) + 1))
    scheme = int(rng.integers(0, 2))
    max_cuckoo = int(rng.choice([0, 1, 2, 5, 6]))
    g = Geometry(mem_p=mem_p, scheme=scheme, iblock_p=iblock_p, max_cuckoo=max_cuckoo)
    og = ogeom(g)
    L.mkv_set_tuning(b"tail_rounds", int(rng.choice([0, 1, 3, 48])))
    L.mkv_set_tuning(b"insert_chunk", int(rng.choice([1024, 2048, 8192, 65536])))
    rng2 = np.random.default_rng(5000 + seed)        # (as tests/test_gpu_fuzz.py::_run draws them)
    L.mkv_set_tuning(b"tail_groups", int(rng2.choice([1, 2, 5, 32])))
    L.mkv_set_tuning(b"grid_min", int(rng2.choice([0, 0, 64, 2048])))
    L.mkv_set_tuning(b"calm_factor", int(rng2.choice([1, 4])))
    rng3 = np.random.default_rng(11000 + seed)
    L.mkv_set_tuning(b"debug_skip", int(rng3.choice([0, 0, 1, 2])) + int(rng3.choice([0, 0, 0, 4])))
    L.mkv_set_tuning(b"round_blocks", int(rng3.choice([1, 3, 64, 512, 512])))
    for k, v in over.items():
        L.mkv_set_tuning(k.encode(), int(v))
    shape = str(np.random.default_rng(7000 + seed).choice(["random", "random", "random", "ref", "cluster"]))
    fill = float(rng.choice([0.0, 0.3, 0.6, 0.85, 0.97]))
    sig_space = int(rng.choice([40, 5000, 1 << 20, 1 << 31]))
    exp = random_table(g, rng, fill, sig_space)
    idx = HashIndex(g, lib=L)
    idx.load_table(exp)
    nslots = g.nbuckets * 8
    print(dict(mem_p=mem_p, iblock_p=iblock_p, scheme=scheme, max_cuckoo=max_cuckoo, fill=fill, sig_space=sig_space))
    for step in range(3):
        op = rng.choice(["insert", "insert", "delete", "search"])
        n = int(rng.integers(1, max(2, min(40000, nslots))))
        e = random_requests(g, rng, n, sig_space, hot=float(rng.choice([0.0, 0.1, 0.6])),
                            zero_low=float(rng.choice([0.0, 0.1])), zeros=float(rng.choice([0.0, 0.03])), shape=shape)
        if op == "insert":
            how = str(rng.choice(["single", "region", "app", "uneven"]))
            if iblock_p == 0 and how == "app":
                how = "single"
            blocks = split_blocks(e, g, rng, how)
            before = exp.copy()
            O.insert(og, exp, blocks)
            if hasattr(L, "mkv_debug_trace"):
                L.mkv_debug_trace(None, 0, 1)
            idx.insert([dev(b.reshape(-1, 3)) for b in blocks], hint=None if rng.random() < 0.7 else 0)
            got = idx.table_numpy()
            print("step", step, op, n, how, idx.last_update_info())
            if not np.array_equal(got, exp):
                bad = np.argwhere((got != exp).any(axis=1))[:, 0]
                print("differing buckets", bad[:10], "of", len(bad))
                allreq = np.concatenate([b.reshape(-1, 3) for b in blocks if len(b)])
                hm, bm = g.hash_mask, g.block_hash_mask
                P = allreq[:, 1] & hm
                A = (((allreq[:, 1] ^ allreq[:, 0]) & bm) | (allreq[:, 1] & ~np.uint32(bm))) & hm
                # where did foreign signatures in the differing bucket come from?
                for B in bad[:1]:
                    for sg in set(int(x) for x in exp[B][:8]) - set(int(x) for x in before[B][:8]) - {0}:
                        src = np.argwhere(before[:, :8] == sg)
                        print("   sig", sg, "was at (bucket, slot)", src.tolist(), "before; requests with that sig:",
                              np.flatnonzero(allreq[:, 0] == sg).tolist())
                        for (b0, s0) in src.tolist():
                            for i in np.argwhere((P == b0) | (A == b0))[:, 0]:
                                print("      touches origin bucket", b0, ": idx", i, "sig", allreq[i, 0], "sig&7", allreq[i, 0] & 7, "P", P[i], "A", A[i])
                if hasattr(L, "mkv_debug_trace"):
                    import ctypes
                    cap = 4 * (1 << 17) + 4
                    buf = (ctypes.c_uint32 * cap)()
                    L.mkv_debug_trace(buf, cap, 0)
                    ev = np.frombuffer(buf, dtype=np.uint32)
                    ne = min(int(ev[0]), 1 << 17)
                    ev = ev[4:4 + 4 * ne].reshape(-1, 4)
                    nb = g.nbuckets
                    per = nb >> iblock_p
                    reg_of = lambda i: int(P[i]) // per
                    R = int(bad[0]) // per
                    names = {1: "round-commit", 2: "replay-commit", 3: "VIOLATION", 4: "take-back", 5: "roll-back",
                             6: "trunc-note", 7: "revive(cut)", 8: "cut-done", 9: "revive(gather)"}
                    print("trace:", ne, "events; region", R, "of the differing bucket:")
                    for code, a, b, c in ev.tolist():
                        if a < len(allreq) and reg_of(a) == R:
                            print("   %-14s idx %d  (%d, %d)  sig %d P %d A %d w %s" % (names.get(code, code), a, b, c, allreq[a, 0], P[a], A[a], ""))
                for B in bad[:3]:
                    print("bucket", B, "\n before", before[B], "\n oracle", exp[B], "\n gpu   ", got[B])
                    for i in np.argwhere((P == B) | (A == B))[:, 0]:
                        print("   idx", i, "sig", allreq[i, 0], "P", P[i], "A", A[i], "loc", allreq[i, 2],
                              "| P row before", before[P[i]][:8], "| A row before", before[A[i]][:8])
                return
        elif op == "delete":
            occ = np.argwhere(exp[:, :8] != 0)
            if len(occ):
                pick = occ[rng.integers(0, len(occ), size=n // 2)]
                e[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
                e[: len(pick), 1] = pick[:, 0]
                e[: len(pick), 2] = exp[pick[:, 0], 8 + pick[:, 1]]
            O.delete(og, exp, e)
            idx.delete(dev(e))
        else:
            rng.integers(0, 2)
    print("no mismatch")


if __name__ == "__main__":
    main()
