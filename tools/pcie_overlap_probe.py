#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rate of the default workload with copies and kernels
overlapped: one stream copies the requests of batch i+1 in, one runs the cycle of batch i,
one copies the results of batch i-1 out (three rotating device buffer sets, events between
the streams).  usage: pcie_overlap_probe.py [steps]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import megakv_amd as M  # noqa: E402
from megakv_amd.index import HashIndex  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    args = argparse.Namespace(keying="mix", lf=0.2, workload="zipf95", batch_log2=20, sets="new", mem_p=30,
                              memset=False, overlap="fused", graph=False)
    sg = bench.Stream(args, 0, 1, 30)
    eng = bench.SingleGpu(HashIndex(M.Geometry(mem_p=30), device="cuda:0"), args)
    eng.preload(sg.P, sg.keying)
    nset = 3
    dev = [eng.stage(*sg.batch()) for _ in range(nset)]
    host = []
    for _ in range(steps):
        b = eng.stage(*sg.batch())
        host.append({"q": b["q"].cpu().pin_memory(), "sets": b["sets_all"].cpu().pin_memory(),
                     "out": torch.empty_like(b["out"], device="cpu").pin_memory()})
    torch.cuda.synchronize()
    s_in, s_run, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    eng.reserve(s_run)
    ev_in = [torch.cuda.Event() for _ in range(nset)]
    ev_run = [torch.cuda.Event() for _ in range(nset)]
    ev_out = [torch.cuda.Event() for _ in range(nset)]

    def run(hs):
        for i, h in enumerate(hs):
            k = i % nset
            b = dev[k]
            with torch.cuda.stream(s_in):
                if i >= nset:
                    s_in.wait_event(ev_run[k])          # the set's previous cycle has read its requests
                b["q"].copy_(h["q"], non_blocking=True)
                b["sets_all"].copy_(h["sets"], non_blocking=True)
                ev_in[k].record(s_in)
            with torch.cuda.stream(s_run):
                s_run.wait_event(ev_in[k])
                if i >= nset:
                    s_run.wait_event(ev_out[k])         # ... and its results have been copied out
                eng._launch(b, s_run, None)
                ev_run[k].record(s_run)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_run[k])
                h["out"].copy_(b["out"], non_blocking=True)
                ev_out[k].record(s_out)
        torch.cuda.synchronize()

    run(host[:6])
    t0 = time.perf_counter()
    run(host)
    dt = time.perf_counter() - t0
    hit = float(((host[-1]["out"][:, 0] != 0) | (host[-1]["out"][:, 1] != 0)).float().mean())
    print(f"overlapped: {steps} steps of 2^20 requests, {dt / steps * 1e6:.1f} us/step = {steps * (1 << 20) / dt / 1e6:.1f} Mops "
          f"(H2D {host[0]['q'].numel() * 4 + host[0]['sets'].numel() * 4} B + D2H {host[0]['out'].numel() * 4} B per step), hit rate {hit:.4f}")


if __name__ == "__main__":
    main()
