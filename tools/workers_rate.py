#!/usr/bin/env python3
"""Device-resident rate of one scheduler cycle for W workers at the reference's batch sizes
(src/mega.c:136,143-144: <= 32 767 searches, 8 x <= 4 096 inserts, < 4 096 deletes per worker):
the per-worker calls (memset + gpu_hash_search + gpu_hash_delete + gpu_hash_insert on a stream per
worker, one device synchronisation per cycle -- src/mega_scheduler.c:393-504) against ONE
mkv_cycle_workers call, requests resident in HBM, HIP events around the cycles.
usage: workers_rate.py [workers=16] [cycles=40] [searches=7000] [inserts=6000] [deletes=1000]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import megakv_amd as M  # noqa: E402
from megakv_amd.index import _dev_ptr, _stream_ptr  # noqa: E402
from megakv_amd.lib import MkvWorker  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ns, ni, nd = (int(sys.argv[k]) if len(sys.argv) > k else v for k, v in ((3, 7000), (4, 6000), (5, 1000)))
    g = M.Geometry(mem_p=30)
    idx = M.HashIndex(g)
    rng = np.random.default_rng(5)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
    streams = [torch.cuda.Stream() for _ in range(W)]
    work = []                                   # per cycle, per worker: staged device buffers
    key = 1
    for c in range(cycles):
        row = []
        for w in range(W):
            k = np.arange(key, key + ni, dtype=np.uint64)
            key += ni
            z = (k * np.uint64(0x9E3779B97F4A7C15)) ^ (k >> np.uint64(29))
            z = (z ^ (z >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93) & np.uint64(0xFFFFFFFFFFFFFFFF)
            sig = (z & np.uint64(0xffffffff)).astype(np.uint32)
            sig[sig == 0] = 1
            hsh = (z >> np.uint64(32)).astype(np.uint32)
            e = np.stack([sig, hsh, k.astype(np.uint32)], axis=1)
            region = (hsh >> np.uint32(29)).astype(np.int64)          # the receiver's binning, src/mega_recv.c:476-477
            blocks = [dev(e[region == b].reshape(-1, 3)) for b in range(8)]
            ptrs, cnts, keep = idx.stage_blocks(blocks)
            q = dev(e[rng.integers(0, ni, size=ns), :2])
            out = torch.empty((ns, 2), dtype=torch.int32, device="cuda")
            d = dev(e[:nd]) if nd else None
            row.append((q, out, d, ptrs, cnts, keep))
        work.append(row)
    torch.cuda.synchronize()
    L = idx.lib

    def per_worker(c):
        for w in range(W):
            q, out, d, ptrs, cnts, keep = work[c][w]
            s = _stream_ptr(streams[w])
            out.zero_() if False else None
            L.mkv_cycle(ctypes.byref(idx._g), _dev_ptr(idx.table), _dev_ptr(q), _dev_ptr(out), ns,
                        _dev_ptr(d) if d is not None else None, nd, _dev_ptr(ptrs), _dev_ptr(cnts), 8, 0,
                        0x100 | 0x200, s, None, 0)          # MKV_CYCLE_MEMSET | MKV_CYCLE_NO_OVERLAP: the three-call sequence
        torch.cuda.synchronize()

    def one_call(c):
        ws = (MkvWorker * W)()
        for w in range(W):
            q, out, d, ptrs, cnts, keep = work[c][w]
            ws[w].search_in, ws[w].search_out, ws[w].num_search = _dev_ptr(q).value, _dev_ptr(out).value, ns
            ws[w].delete_in, ws[w].num_delete = (_dev_ptr(d).value if d is not None else None), nd
            ws[w].blk_input, ws[w].blk_elem_num, ws[w].num_blks = _dev_ptr(ptrs).value, _dev_ptr(cnts).value, 8
        rc = L.mkv_cycle_workers(ctypes.byref(idx._g), _dev_ptr(idx.table), ws, W, _stream_ptr(streams[0]))
        assert rc == 0
        torch.cuda.synchronize()

    ops = W * (ns + ni + nd)
    for name, fn in (("per-worker calls, a stream each", per_worker), ("one mkv_cycle_workers call", one_call)):
        idx.table.zero_()
        torch.cuda.synchronize()
        for c in range(3):
            fn(c)
        t0 = time.perf_counter()
        for c in range(3, cycles):
            fn(c)
        dt = time.perf_counter() - t0
        print(f"{W} workers x ({ns} searches, {ni} inserts, {nd} deletes), {cycles - 3} cycles, {name}: "
              f"{dt / (cycles - 3) * 1e6:.1f} us per cycle (one synchronisation each), {ops * (cycles - 3) / dt / 1e6:.0f} Mops")


if __name__ == "__main__":
    main()
