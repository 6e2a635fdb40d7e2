#!/usr/bin/env python3
"""Runs ONLY the search kernel (for rocprofv3 --pmc passes): pre-load, then
`reps` launches over distinct 2^20-request GET batches.
usage: search_only.py [uniform|zipf] [mix|ref] [reps]"""
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
    theta = 0.0 if (len(sys.argv) < 2 or sys.argv[1] == "uniform") else 0.99
    keying = W.KEYING_MIX if (len(sys.argv) < 3 or sys.argv[2] == "mix") else W.KEYING_REF
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    idx = M.HashIndex(M.Geometry(mem_p=30))
    P = W.preload_count(30, 0.2)
    for lo in range(1, P, 1 << 20):
        k = np.arange(lo, min(lo + (1 << 20), P), dtype=np.uint32)
        idx.insert([dev(b) for b in W.bin_inserts(W.make_ielem(k, None, keying), 30, 3, W.BIN_REGION)])
    z = W.ZipfKeys(P - 2, theta, 21)
    n = 1 << 20
    qs = [dev(W.make_selem(z.keys(n), keying)) for _ in range(reps)]
    out = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for q in qs:
        idx.search(q, out=out, zero_miss=False)
    torch.cuda.synchronize()
    print("done", reps)


if __name__ == "__main__":
    main()
