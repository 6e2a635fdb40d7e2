#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rate of the default workload WITHOUT copies: the requests stay in
pinned host memory, which the device maps at the same address; the cycle's kernels read them from
there and write the results back there, so transfer and probing overlap inside the kernels (the
reference's scheduler copies: src/mega_scheduler.c:404-419,484-494).  Variants: everything in host
memory / only the results / only the requests.  Results are compared with the copying path.
usage: zero_copy_probe.py [steps]"""
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
    idx = HashIndex(M.Geometry(mem_p=30), device="cuda:0")
    eng = bench.SingleGpu(idx, args)
    eng.preload(sg.P, sg.keying)
    n = 1 << 20
    batches = []
    for _ in range(steps + 4):
        b = eng.stage(*sg.batch())
        h = {"q": b["q"].cpu().pin_memory(), "out": torch.zeros_like(b["out"], device="cpu").pin_memory(),
             "sets_all": b["sets_all"].cpu().pin_memory()}
        offs = np.concatenate([[0], np.cumsum([k.shape[0] for k in b["keep"]])])
        blocks = [h["sets_all"][offs[i]:offs[i + 1]] for i in range(len(b["keep"]))]
        h["ptrs"], h["cnts"], h["keep"] = idx.stage_blocks(blocks)     # pointers into the pinned buffer
        h["n_set"] = b["n_set"]
        batches.append((b, h))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    eng.reserve(stream)
    # the table changes with every cycle (SETs): each variant runs the same batches in the same
    # order on a table restored to the same image, and must produce the same results
    image = idx.table.clone()
    ref = None
    for name, host_in, host_out, copies in (
            ("copies (H2D, cycle, D2H on one stream)", False, False, True),
            ("requests and results in pinned host memory", True, True, True),
            ("results in host memory, requests copied", False, True, True),
            ("requests in host memory, results copied", True, False, True),
            ("  for scale: results to host, requests resident", False, True, False),
            ("  for scale: requests from host, results stay", True, False, False),
            ("  for scale: nothing crosses PCIe", False, False, False)):
        idx.table.copy_(image)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            t0 = None
            for i, (b, h) in enumerate(batches):
                if i == 4:
                    stream.synchronize()
                    t0 = time.perf_counter()
                if not host_in and copies:
                    b["q"].copy_(h["q"], non_blocking=True)
                    b["sets_all"].copy_(h["sets_all"], non_blocking=True)
                src = h if host_in else b
                out = h["out"] if host_out else b["out"]
                idx.cycle(src["q"], out, src["ptrs"], src["cnts"], len(src["keep"]), stream=stream,
                          insert_hint=src["n_set"])
                if not host_out and copies:
                    h["out"].copy_(b["out"], non_blocking=True)
            stream.synchronize()
            dt = time.perf_counter() - t0
        if not host_out and not copies:
            for b, h in batches:
                h["out"].copy_(b["out"])
        got = np.stack([h["out"].numpy().copy() for _, h in batches])
        table = idx.table.cpu()
        if ref is None:
            ref = (got, table)
        same = bool(np.array_equal(got, ref[0]) and torch.equal(table, ref[1]))
        print(f"{name:48s}: {dt / steps * 1e6:8.1f} us/step = {steps * n / dt / 1e6:8.1f} Mops   "
              f"results and table equal to the copying path: {same}", flush=True)
        for _, h in batches:
            h["out"].zero_()
    # requests copied in on a second stream (two device buffer sets), results written to host
    # memory by the kernels: the copy of batch i+1 runs beside the cycle of batch i
    idx.table.copy_(image)
    torch.cuda.synchronize()
    s_copy = torch.cuda.Stream()
    sets = [batches[0][0], batches[1][0]]
    ev_copy = [torch.cuda.Event(), torch.cuda.Event()]
    ev_run = [torch.cuda.Event(), torch.cuda.Event()]
    t0 = None
    for i, (_, h) in enumerate(batches):
        if i == 4:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        k = i & 1
        d = sets[k]
        with torch.cuda.stream(s_copy):
            if i >= 2:
                s_copy.wait_event(ev_run[k])
            d["q"].copy_(h["q"], non_blocking=True)
            d["sets_all"].copy_(h["sets_all"], non_blocking=True)
            ev_copy[k].record(s_copy)
        with torch.cuda.stream(stream):
            stream.wait_event(ev_copy[k])
            idx.cycle(d["q"], h["out"], d["ptrs"], d["cnts"], len(d["keep"]), stream=stream, insert_hint=h["n_set"])
            ev_run[k].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = np.stack([h["out"].numpy().copy() for _, h in batches])
    same = bool(np.array_equal(got, ref[0]) and torch.equal(idx.table.cpu(), ref[1]))
    print(f"{'requests copied on a 2nd stream, results to host':48s}: {dt / steps * 1e6:8.1f} us/step = "
          f"{steps * n / dt / 1e6:8.1f} Mops   results and table equal to the copying path: {same}", flush=True)


if __name__ == "__main__":
    main()
