#!/usr/bin/env python3
"""Per-call diagnostics of the insert path: pre-load to a load factor, then time
insert batches of a given size and print the library's update info.
usage: insert_probe.py <lf> <batch> <nbatches> [new|upd] [cuckoo|2choice] [tuning=value ...]"""
import os
import sys
import time

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
    sets = sys.argv[4] if len(sys.argv) > 4 else "new"
    scheme = M.SCHEME_2CHOICE if (len(sys.argv) > 5 and sys.argv[5] == "2choice") else M.SCHEME_CUCKOO
    g = M.Geometry(mem_p=30, scheme=scheme)
    for a in sys.argv[4:]:
        if "=" in a:
            name, value = a.split("=")
            assert M.lib.load_library(diagnostics=True).mkv_set_tuning(name.encode(), int(value)) == 0, a
    idx = M.HashIndex(g, lib=M.lib.load_library(diagnostics=True))
    idx.reserve(max(batch, 1 << 20) + 8)
    P = W.preload_count(30, lf)
    t0 = time.time()
    for lo in range(1, P, 1 << 20):
        k = np.arange(lo, min(lo + (1 << 20), P), dtype=np.uint32)
        idx.insert([dev(b) for b in W.bin_inserts(W.make_ielem(k, None, W.KEYING_MIX), 30, 3, W.BIN_REGION)])
    torch.cuda.synchronize()
    print(f"preload {P} keys: {time.time() - t0:.2f}s  last info {idx.last_update_info()}", flush=True)
    z = W.ZipfKeys(P - 2, 0.99, 77)
    nxt = P
    for i in range(nb):
        if sets == "new":
            k = np.arange(nxt, nxt + batch, dtype=np.uint64).astype(np.uint32)
            nxt += batch
        else:
            k = z.keys(batch)
        blocks = [dev(b) for b in W.bin_inserts(W.make_ielem(k, None, W.KEYING_MIX), 30, 3, W.BIN_REGION)]
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        idx.insert(blocks)
        b.record()
        torch.cuda.synchronize()
        try:
            big = idx.sorted_stats()
        except Exception:                       # (an earlier round's build of the library, tools/ab_lines.sh)
            big = None
        print(f"batch {i}: {a.elapsed_time(b) * 1e3:9.1f} us  {idx.last_update_info()} tail phases us {idx.tail_phases_us()} "
              f"big passes {big}", flush=True)


if __name__ == "__main__":
    main()
