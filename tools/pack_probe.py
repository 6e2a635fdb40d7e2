#!/usr/bin/env python3
"""Times mkv_pack_requests and mkv_unpack_results alone (events around 50 calls).
usage: pack_probe.py [G ...]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megakv_amd.lib import load_library
from megakv_amd.index import _dev_ptr, _stream_ptr
L = load_library(diagnostics=True)      # (the diagnostics build: make -C megakv_amd/csrc DIAG=1)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(g, n_get, n_set, three, threads=1024):
    L.mkv_set_tuning(b"pack_three_pass", three)
    L.mkv_set_tuning(b"pack_threads", threads)
    rng = np.random.default_rng(1)
    mem_p_agg = 30 + (g.bit_length() - 1)
    mask, shift = (1 << (mem_p_agg - 6)) - 1, 24
    gets = torch.from_numpy(rng.integers(0, 2**32, size=(max(n_get, 1), 2), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    sets = torch.from_numpy(rng.integers(0, 2**32, size=(max(n_set, 1), 3), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    cap_get = (int(n_get / g * 1.25) + 255) // 256 * 256
    cap_set = (int(n_set / g * 1.25) + 15) // 16 * 16 + 16
    sw = int(L.mkv_pack_slot_words(cap_get, cap_set))
    sendbuf = torch.zeros((g, sw), dtype=torch.int32, device="cuda")
    perm = torch.zeros((max(n_get, 1),), dtype=torch.int32, device="cuda")
    nbytes = int(L.mkv_pack_scratch_bytes(max(n_get, 1), max(n_set, 1), g))
    scratch = torch.zeros((nbytes // 4 + 16,), dtype=torch.int32, device="cuda")
    overflow = torch.zeros((1,), dtype=torch.int32, device="cuda")
    back = torch.zeros((g * max(cap_get, 1), 2), dtype=torch.int32, device="cuda")
    out = torch.zeros((max(n_get, 1), 2), dtype=torch.int32, device="cuda")

    def pack():
        L.mkv_pack_requests(_dev_ptr(gets), n_get, _dev_ptr(sets), n_set, mask, shift, g, _dev_ptr(sendbuf), cap_get,
                            cap_set, _dev_ptr(perm), _dev_ptr(scratch), scratch.numel() * 4, _dev_ptr(overflow), _stream_ptr(None))
    t_pack = timed(pack)
    t_un = timed(lambda: L.mkv_unpack_results(_dev_ptr(back), _dev_ptr(perm), n_get, _dev_ptr(out), _stream_ptr(None))) if n_get else 0.0
    print(f"G={g} n_get={n_get} n_set={n_set} three_pass={three} threads={threads}: pack {t_pack:.1f} us  un-pack {t_un:.1f} us  "
          f"overflow {int(overflow.item())}")


for g in ([int(a) for a in sys.argv[1:]] or [1, 2, 8]):
    for th in (1024, 512, 256):
        run(g, 996148, 52428, 0, th)
        run(g, 524288, 524288, 0, th)
    run(g, 996148, 0, 0)
    run(g, 0, 52428, 0)
    run(g, 524288, 524288, 0)
    run(g, 996148, 52428, 1)
