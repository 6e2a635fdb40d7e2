#!/usr/bin/env python3
"""The W-rank sharded step on ONE GPU with every peer looped back to this process (the exchanges are
device copies of the same sizes): full-size batches (2^20 requests, 95/5) through the native engine's
default schedule, microseconds per batch, and -- MKV_SHARD_TRACE=n in the environment -- the library's
device-timestamp table of n batches near the end of the run (stderr).
usage: loopback_trace.py <world> [batches, default 64] [mem_p of the local table, default 30] [uniform|zipf]
zipf: bench.py's default workload for rank 0 of <world> ranks (aggregate table pre-loaded to load factor 0.2 -- this
rank's shard of it --, 95/5 GET/SET, Zipf-0.99 GETs, fresh SET keys) instead of uniform keys on an empty table: what
BASELINE.md's predicted scaling curve is made from (MKV_SHARD_FAKE_XFER_US stands for the wire)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import late_imports, sharded_engine as ShardedIndex  # noqa: E402


def main():
    world = int(sys.argv[1])
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    mem_p = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    zipf = len(sys.argv) > 4 and sys.argv[4] == "zipf"
    late_imports()
    n = 1 << 20
    n_set = n // 20
    n_get = n - n_set
    gen = None
    if zipf:
        import argparse
        import bench
        import workload as W
        a = argparse.Namespace(keying="mix", lf=0.2, workload="zipf95", batch_log2=20, sets="new", sorted_gets=False)
        gen = bench.Stream(a, 0, world, mem_p + world.bit_length() - 1)
        n_get, n_set = gen.n_get, gen.n_set
    # (a hot key sends all its GETs to one shard: slots sized like bench.py sizes them)
    from megakv_amd.lib import load_library
    sh = ShardedIndex(mem_p, world, 0, 0, "cuda:0", n_get, n_set, comm=None, cap_factor=1.6 if zipf and world > 2 else 1.3,
                      lib=load_library(diagnostics=True))     # (the looped-back peers live in the diagnostics build)
    sh.enable_native(None, loopback=True)
    rng = np.random.default_rng(7)
    try:
        batches = []
        if zipf:
            sh.preload(gen.P, gen.keying)            # (this rank's shard of the aggregate pre-load)
        for step in range(nb if not zipf else 0):
            z = rng.integers(1, 2**63, size=n, dtype=np.uint64)
            sig = (z & np.uint64(0xffffffff)).astype(np.uint32)
            sig[sig == 0] = 1
            hsh = (z >> np.uint64(32)).astype(np.uint32)
            gets = np.ascontiguousarray(np.stack([sig[:n_get], hsh[:n_get]], axis=1))
            sets = np.ascontiguousarray(np.stack([sig[n_get:], hsh[n_get:], np.arange(1, n_set + 1, dtype=np.uint32) + step], axis=1))
            batches.append(sh.stage(gets, sets))
        for step in range(nb if zipf else 0):
            gets, sets = gen.batch()
            batches.append(sh.stage(gets, sets))
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sh.run_pipelined(batches)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"world {world} loopback, run {rep}: {dt / nb * 1e6:.1f} us per batch of 2^20 "
                  f"({n * nb / dt / 1e9:.2f} Gops through this rank), overflow {sh.overflowed()}")
    finally:
        sh.native.close()


if __name__ == "__main__":
    main()
