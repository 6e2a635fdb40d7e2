#!/usr/bin/env python3
"""Kernel-only rates of insert and delete on the pre-loaded MEM_P=30 table (LF 0.2, K-mix):
HIP events around one call, median over distinct batches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megakv_amd as M
import workload as W

def dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()

def main():
    for scheme, name in ((M.SCHEME_CUCKOO, "cuckoo"), (M.SCHEME_2CHOICE, "2choice")):
        idx = M.HashIndex(M.Geometry(mem_p=30, scheme=scheme)); idx.reserve((1 << 20) + 8)
        P = W.preload_count(30, 0.2)
        for lo in range(1, P, 1 << 20):
            k = np.arange(lo, min(lo + (1 << 20), P), dtype=np.uint32)
            idx.insert([dev(b) for b in W.bin_inserts(W.make_ielem(k, None, W.KEYING_MIX), 30, 3, W.BIN_REGION)])
        torch.cuda.synchronize()
        nxt = P
        for n in (32768, 52428, 1 << 20):
            ti, td = [], []
            for rep in range(12):
                k = np.arange(nxt, nxt + n, dtype=np.uint64).astype(np.uint32); nxt += n
                e = W.make_ielem(k, None, W.KEYING_MIX)
                blocks = [dev(b) for b in W.bin_inserts(e, 30, 3, W.BIN_REGION)]
                de = dev(e)
                ptrs, cnts, keep = idx.stage_blocks(blocks)
                torch.cuda.synchronize()
                a, b, c, d = (torch.cuda.Event(enable_timing=True) for _ in range(4))
                a.record(); idx.insert_staged(ptrs, cnts, len(keep), None, n); b.record()
                c.record(); idx.delete(de); d.record()
                torch.cuda.synchronize()
                ti.append(a.elapsed_time(b) * 1e3); td.append(c.elapsed_time(d) * 1e3)
            mi, md = float(np.median(ti[2:])), float(np.median(td[2:]))
            print(f"{name} n={n}: insert {mi:.1f} us = {n / mi:.0f} Mops   delete {md:.1f} us = {n / md:.0f} Mops", flush=True)

if __name__ == "__main__":
    main()
