#!/usr/bin/env python3
"""What the memory system gives for random reads (no hashing, no compare): pieces of
64 / 128 / 256 bytes at pseudo-random places of a 1 GiB (and an L2-sized) buffer, issued like
the search kernel issues its bucket loads.  The 64-byte / 1 GiB line is the ceiling of a kernel
whose every request is two random buckets.  usage: random_read_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megakv_amd.lib import load_library  # noqa: E402
from megakv_amd.index import _dev_ptr, _stream_ptr  # noqa: E402

L = load_library(diagnostics=True)      # (the diagnostics build: make -C megakv_amd/csrc DIAG=1)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for mib in (1024, 16):
    buf = torch.empty(mib << 18, dtype=torch.int32, device="cuda").random_()
    for gran in (32, 64, 128, 256):
        for groups in (1 << 18, 1 << 20):
            pieces = (1 << 23) // groups * 4 if gran == 64 else (1 << 23) // groups * 2
            pieces = max(8, pieces & ~7)
            best = None
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                n = L.mkv_debug_random_read(_dev_ptr(buf), mib << 20, gran, groups, pieces, _dev_ptr(sink), _stream_ptr(None))
                b.record()
                torch.cuda.synchronize()
                assert n > 0, n
                t = a.elapsed_time(b) * 1e-3
                best = t if best is None or t < best else best
            print(f"buffer {mib:5d} MiB  piece {gran:3d} B  groups {groups:8d} x {pieces:3d} pieces: "
                  f"{n / 1e6:8.1f} MB in {best * 1e6:7.1f} us = {n / best / 1e12:5.2f} TB/s ({n / best / 8e12:4.2f} of 8 TB/s)")

# at the size of ONE search launch: 2^21 bucket reads (2^20 GETs) and multiples
buf = torch.empty(1024 << 18, dtype=torch.int32, device="cuda").random_()
for groups, pieces in ((1 << 18, 8), (1 << 19, 8), (1 << 17, 16), (1 << 18, 16), (1 << 18, 32)):
    best = None
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = L.mkv_debug_random_read(_dev_ptr(buf), 1024 << 20, 64, groups, pieces, _dev_ptr(sink), _stream_ptr(None))
        b.record()
        torch.cuda.synchronize()
        assert n > 0, n
        t = a.elapsed_time(b) * 1e-3
        best = t if best is None or t < best else best
    print(f"groups {groups} x {pieces} pieces of 64 B = {groups * pieces / 2 ** 20:.1f} M pieces: {best * 1e6:.1f} us = "
          f"{groups * pieces / best / 1e9:.1f} G pieces/s")
