#!/usr/bin/env python3
"""Who loses round 1 in a bench.py step (DIAGNOSTICS build): for every request round 2 or the tail committed, the
requests of the pass that share its primary bucket -- as their primary or as their alternate bucket -- and how they
were registered.  usage: loser_probe.py [zipf95|zipf50] [new|upd] [steps, default 12] [lf, default 0.2] [tuning=value ...]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import megakv_amd as M  # noqa: E402
from megakv_amd.index import HashIndex  # noqa: E402
from megakv_amd.lib import load_library  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if "=" not in a]
    tunings = [a.split("=") for a in sys.argv[1:] if "=" in a]
    wl = argv[0] if len(argv) > 0 else "zipf95"
    sets = argv[1] if len(argv) > 1 else "new"
    steps = int(argv[2]) if len(argv) > 2 else 12
    lf = float(argv[3]) if len(argv) > 3 else 0.2
    bench.late_imports()
    args = argparse.Namespace(keying="mix", lf=lf, workload=wl, batch_log2=20, sets=sets, mem_p=30, memset=False,
                              overlap="fused", graph=False, sorted_gets=False)
    sg = bench.Stream(args, 0, 1, 30)
    L = load_library(diagnostics=True)
    L.mkv_debug_dump.restype = ctypes.c_int
    L.mkv_debug_dump.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    for name, value in tunings:
        assert L.mkv_set_tuning(name.encode(), int(value)) == 0, name
    g = M.Geometry(mem_p=30)
    eng = bench.SingleGpu(HashIndex(g, device="cuda:0", lib=L), args)
    eng.preload(sg.P, sg.keying)
    s = torch.cuda.Stream()
    eng.reserve(s)
    batches = [eng.stage(*sg.batch()) for _ in range(steps)]
    torch.cuda.synchronize()
    hm, bm = g.hash_mask, (1 << (30 - 6 - 3)) - 1
    for i, b in enumerate(batches):
        eng.step(b, s, None)
        s.synchronize()
        info = eng.idx.last_update_info(s, cycle_insert=True)
        n = info["total"]
        st = np.zeros(n, np.uint8)
        rc = np.zeros((n, 4), np.uint32)
        assert L.mkv_debug_dump(1, st.ctypes.data, rc.ctypes.data, n) == 0
        sig, h, aux = rc[:, 0], rc[:, 1], rc[:, 3]
        P = h & hm
        A = (((h ^ sig) & bm) | (h & ~np.uint32(bm))) & hm
        late = np.nonzero((st != 1) & (st != 2) & (st != 6))[0]      # not DONE / ROUND1 / CHAIN1
        print(f"step {i}: total {n} round1 {info['round1']} helped {info['helped']} losers {info['losers']} round2 {info['round2']} "
              f"tail {info['tail_parallel']}+{info['tail_serial']}  late requests {len(late)}")
        for j in late[:6]:
            sameP = np.nonzero(P == P[j])[0]
            asA = np.nonzero((A == P[j]) & (P != P[j]))[0]
            desc = []
            for k in sorted(set(sameP) | set(asA)):
                kind = "P" if P[k] == P[j] else "A"
                desc.append(f"{k}{'*' if k == j else ''}:{kind}{'l' if aux[k] & 1 else ''}{'p' if aux[k] & 16 else ''}:st{st[k]}")
            print(f"   idx {j} tag {st[j]} bucket {P[j]:#x}: " + " ".join(desc))


if __name__ == "__main__":
    main()
