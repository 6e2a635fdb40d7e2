#!/usr/bin/env python3
"""Times the search kernel alone over its tuning knobs (requests in flight per
lane group, non-temporal loads, fused miss zeroing) on uniform and Zipf-0.99
GET batches against the pre-loaded MEM_P=30 table.  GPU box only."""
import json
import sys
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import megakv_amd as M  # noqa: E402
import workload as W  # noqa: E402
from megakv_amd.lib import load_library  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def main():
    keying = W.KEYING_MIX if (len(sys.argv) < 2 or sys.argv[1] == "mix") else W.KEYING_REF
    L = load_library(diagnostics=True)
    g = M.Geometry(mem_p=30)
    idx = M.HashIndex(g, lib=M.lib.load_library(diagnostics=True))
    P = W.preload_count(30, 0.2)
    for lo in range(1, P, 1 << 20):
        k = np.arange(lo, min(lo + (1 << 20), P), dtype=np.uint32)
        blocks = W.bin_inserts(W.make_ielem(k, None, keying), 30, 3, W.BIN_REGION)
        idx.insert([dev(b) for b in blocks])
    torch.cuda.synchronize()
    n = 1 << 20
    nb = 24
    res = []
    for theta in (0.0, 0.99):
        z = W.ZipfKeys(P - 2, theta, 21)
        qs = [dev(W.make_selem(z.keys(n), keying)) for _ in range(nb)]
        out = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
        for u in (1, 2, 4, 8):
            for nt in (0, 1):
                for zero in (False, True):
                    L.mkv_set_tuning(b"search_u", u)
                    L.mkv_set_tuning(b"search_nt", nt)
                    for q in qs[:4]:
                        idx.search(q, out=out, zero_miss=zero)
                    torch.cuda.synchronize()
                    evs = []
                    for rep in range(2):
                        for q in qs:
                            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            a.record()
                            idx.search(q, out=out, zero_miss=zero)
                            b.record()
                            evs.append((a, b))
                    torch.cuda.synchronize()
                    us = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
                    r = {"theta": theta, "u": u, "nt": nt, "zero_miss": zero, "avg_us": round(float(us.mean()), 2),
                         "p10": round(float(np.percentile(us, 10)), 2), "GBps_algo": round(144 * n / us.mean() / 1e3, 1)}
                    res.append(r)
                    print(json.dumps(r), flush=True)
        # the alternative layouts at their best setting: 8 lanes per request, and the LDS-staged
        # ballot probe of the reference's structure
        L.mkv_set_tuning(b"search_u", 4)
        L.mkv_set_tuning(b"search_nt", 0)
        for name in (b"search_wide", b"search_lds"):
            L.mkv_set_tuning(name, 1)
            for q in qs[:4]:
                idx.search(q, out=out, zero_miss=True)
            torch.cuda.synchronize()
            evs = []
            for rep in range(2):
                for q in qs:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    idx.search(q, out=out, zero_miss=True)
                    b.record()
                    evs.append((a, b))
            torch.cuda.synchronize()
            us = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
            print(json.dumps({"theta": theta, "layout": name.decode(), "avg_us": round(float(us.mean()), 2),
                              "p10": round(float(np.percentile(us, 10)), 2),
                              "GBps_algo": round(144 * n / us.mean() / 1e3, 1)}), flush=True)
            L.mkv_set_tuning(name, 0)


if __name__ == "__main__":
    main()
