#!/usr/bin/env python3
"""What the tail launch of a bench.py step spends where: the step's workload through the DIAGNOSTICS
build (per-phase clocks in k_tail), update info and phase times per step.
usage: tail_phases.py [zipf95|zipf50] [new|upd] [steps, default 8] [lf, default 0.2] [tuning=value ...]
phases (us): 0 roll-back, 1 gather, 2 exact replay below a barrier, 3 retag, 4 in-group rounds, 5 serial replay,
6-7 gather details; 8.. inside the grid-wide rounds: register, meeting, decide, meeting, election + meeting,
commit, meeting -- or, when no grid-wide round ran, the diagnostics words: 8 = round 1's slowest work-group (ticks
of 10 ns), 11 = its phase 2 (the queued requests), 15 = the longest queue; 9 / 10 / 12 - 14 = in-group rounds' counters."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import megakv_amd as M  # noqa: E402
from megakv_amd.index import HashIndex  # noqa: E402
from megakv_amd.lib import load_library  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if "=" not in a]
    tunings = [a.split("=") for a in sys.argv[1:] if "=" in a]      # e.g. debug_skip=4 self_help=0
    wl = argv[0] if len(argv) > 0 else "zipf95"
    sets = argv[1] if len(argv) > 1 else "new"
    steps = int(argv[2]) if len(argv) > 2 else 8
    lf = float(argv[3]) if len(argv) > 3 else 0.2
    bench.late_imports()
    args = argparse.Namespace(keying="mix", lf=lf, workload=wl, batch_log2=20, sets=sets, mem_p=30, memset=False,
                              overlap="fused", graph=False, sorted_gets=False)
    sg = bench.Stream(args, 0, 1, 30)
    L = load_library(diagnostics=True)
    for name, value in tunings:
        assert L.mkv_set_tuning(name.encode(), int(value)) == 0, name
    eng = bench.SingleGpu(HashIndex(M.Geometry(mem_p=30), device="cuda:0", lib=L), args)
    eng.preload(sg.P, sg.keying)
    s = torch.cuda.Stream()
    eng.reserve(s)
    batches = [eng.stage(*sg.batch()) for _ in range(steps)]
    torch.cuda.synchronize()
    for i, b in enumerate(batches):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        eng.step(b, s, None)
        z.record(s)
        s.synchronize()
        info = eng.idx.last_update_info(s, cycle_insert=True)
        ph = eng.idx.tail_phases_us(s, cycle_insert=True)
        keep = {k: info[k] for k in ("round1", "round2", "helped", "losers", "tail_parallel", "tail_serial", "undone",
                                     "tail_rounds", "bar_chain", "bar_class")}
        print(f"step {i}: {a.elapsed_time(z) * 1e3:7.1f} us  {keep}  tail phases us {ph}")


if __name__ == "__main__":
    main()
