#!/usr/bin/env python3
"""Which requests seed the order violations of an insert batch: needs the debug build
(make -C megakv_amd/csrc EXTRA=-DMKV_TRACE), whose tail logs what it does.  Pre-loads to a load
factor, inserts batches of Zipf (upd) or fresh (new) keys, prints every violation with the seed's flags.
usage: violation_probe.py <lf> <batch> <nbatches> [new|upd] [cuckoo|2choice]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import megakv_amd as M  # noqa: E402
import workload as W  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def main():
    lf, batch, nb = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sets = sys.argv[4] if len(sys.argv) > 4 else "upd"
    scheme = M.SCHEME_2CHOICE if (len(sys.argv) > 5 and sys.argv[5] == "2choice") else M.SCHEME_CUCKOO
    L = M.lib.load_library(diagnostics=True)
    assert hasattr(L, "mkv_debug_trace"), "build with EXTRA=-DMKV_TRACE"
    g = M.Geometry(mem_p=30, scheme=scheme)
    idx = M.HashIndex(g, lib=M.lib.load_library(diagnostics=True))
    idx.reserve(max(batch, 1 << 20) + 8)
    P = W.preload_count(30, lf)
    for lo in range(1, P, 1 << 20):
        k = np.arange(lo, min(lo + (1 << 20), P), dtype=np.uint32)
        idx.insert([dev(b) for b in W.bin_inserts(W.make_ielem(k, None, W.KEYING_MIX), 30, 3, W.BIN_REGION)])
    torch.cuda.synchronize()
    z = W.ZipfKeys(P - 2, 0.99, 77)
    nxt = P
    cap = 4 * (1 << 17) + 4
    buf = (ctypes.c_uint32 * cap)()
    for i in range(nb):
        if sets == "new":
            k = np.arange(nxt, nxt + batch, dtype=np.uint64).astype(np.uint32)
            nxt += batch
        else:
            k = z.keys(batch)
        blocks = W.bin_inserts(W.make_ielem(k, None, W.KEYING_MIX), 30, 3, W.BIN_REGION)
        allreq = np.concatenate([b.reshape(-1, 3) for b in blocks if len(b)])
        L.mkv_debug_trace(None, 0, 1)
        idx.insert([dev(b) for b in blocks])
        torch.cuda.synchronize()
        L.mkv_debug_trace(buf, cap, 0)
        ev = np.frombuffer(buf, dtype=np.uint32)
        ne = min(int(ev[0]), 1 << 17)
        ev = ev[4:4 + 4 * ne].reshape(-1, 4)
        hm, bm = g.hash_mask, g.block_hash_mask
        print(f"batch {i}: {idx.last_update_info()}  ({ne} trace events)")
        sig_count = {}
        for code, a, b, c in ev.tolist():
            if code != 3:
                continue
            sig, hsh = int(allreq[a, 0]), int(allreq[a, 1])
            Pb = hsh & hm
            Ab = (((hsh ^ sig) & bm) | (hsh & ~bm & 0xffffffff)) & hm
            o = (ctypes.c_uint32 * 9)()
            L.mkv_debug_request(0, int(a), o)
            w = o[4]
            same_key = int(((allreq[:, 0] == sig) & ((allreq[:, 1] & hm) == Pb)).sum())
            on_p = int(((allreq[:, 1] & hm) == b).sum())
            print(f"   violation: idx {a} on bucket {b} ({'P' if b == Pb else 'A' if b == Ab else 'chain'}), flags"
                  f" lazy {w & 1} follower {(w >> 1) & 1} leader {(w >> 2) & 1} checked {(w >> 3) & 1};"
                  f" requests of its key in the batch {same_key}; requests with that bucket as primary {on_p}")


if __name__ == "__main__":
    main()
