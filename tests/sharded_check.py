"""Launches the ranks of a sharded run and checks them against ONE aggregate
table on the CPU oracle (canonical order inside a shard = source rank, then
index; shards are closed regions so their relative order is immaterial)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(world, argv_of, d, env, timeout):
    """Starts the ranks (output to files: a full pipe on one rank would stall every rank at the next
    collective) and waits for ALL of them together: a rank that dies takes the others down at once
    with its log in the failure -- waiting for them one after the other would sit out the whole
    time limit on a survivor blocked in a collective and report nothing of the rank that failed.
    A run whose ranks never got together (the rendezvous port taken between choosing and binding it,
    a rank that could not start) is started once more on another port."""
    import time
    for attempt in range(2):
        port = free_port()
        logs = [open(os.path.join(d, f"rank{r}.log"), "w") for r in range(world)]
        procs = [subprocess.Popen(argv_of(r, port), cwd=ROOT, env=env, stdout=logs[r], stderr=subprocess.STDOUT)
                 for r in range(world)]
        t0 = time.time()
        bad = None
        try:
            while True:
                codes = [p.poll() for p in procs]
                failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
                if failed:
                    bad = failed[0]
                    break
                if all(c == 0 for c in codes):
                    break
                if time.time() - t0 > timeout:
                    bad = -1
                    break
                time.sleep(0.05)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            for f in logs:
                f.close()
        if bad is None:
            return
        tails = {r: open(os.path.join(d, f"rank{r}.log")).read()[-3000:] for r in range(world)}
        early = time.time() - t0 < 120 and not any(os.path.exists(os.path.join(d, f"rank{r}.npz")) for r in range(world))
        if attempt == 0 and early and bad >= 0:
            continue                    # (died before any rank produced anything: once more)
        if bad < 0:
            raise AssertionError(f"ranks still running after {timeout} s:\n" +
                                 "\n".join(f"--- rank {r}\n{t[-800:]}" for r, t in tails.items()))
        raise AssertionError(f"rank {bad} failed:\n{tails[bad]}")


def run_and_check(world, mode, mem_p_local, scheme, seed, steps=4, timeout=600):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
        launch_ranks(world, lambda r, port: [sys.executable, "-m", "tests.sharded_worker", str(r), str(world), str(port),
                                             mode, d, str(mem_p_local), str(scheme), str(seed)], d, env, timeout)
        data = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
        k = world.bit_length() - 1
        og = O.geom(mem_p=mem_p_local + k, scheme=scheme)          # aggregate table, IBLOCK_P = 3
        agg = O.new_table(og)
        for step in range(steps):
            # every rank's GETs see the table before any of this step's SETs
            for r in range(world):
                exp = O.search(og, agg, np.ascontiguousarray(data[r][f"gets{step}"]))
                assert np.array_equal(data[r][f"out{step}"], exp), f"rank {r} step {step}: search results differ"
            O.insert(og, agg, [np.ascontiguousarray(data[r][f"sets{step}"]) for r in range(world)])
        nb_local = 1 << (mem_p_local - 6)
        for r in range(world):
            assert np.array_equal(data[r]["table"], agg[r * nb_local:(r + 1) * nb_local]), f"shard {r} differs"


def run_and_check_zipf(world, mem_p_local, batch_log2, steps, timeout=900, scheme=0):
    """BASELINE configs[4] shape: every rank feeds batches of bench.py's own Stream generator
    (95/5 GET/SET, Zipf-0.99, K-mix keys, fresh SET ids per rank) into the sharded index after
    bench.py's pre-load to load factor 0.2; all ranks share cuda:0, two batches in flight.
    Checked against ONE aggregate oracle table of MEM_P = mem_p_local + log2(world)."""
    import bench
    import workload as W
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MKV_ZIPF_BATCH_LOG2=str(batch_log2),
                   MKV_ZIPF_STEPS=str(steps))
        launch_ranks(world, lambda r, port: [sys.executable, "-m", "tests.sharded_worker", str(r), str(world), str(port),
                                             "gpu-zipf", d, str(mem_p_local), str(scheme), "0"], d, env, timeout)
        data = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    k = world.bit_length() - 1
    mem_p_agg = mem_p_local + k
    og = O.geom(mem_p=mem_p_agg, scheme=scheme)
    agg = O.new_table(og)
    args = bench.parse(["--workload", "zipf95", "--mem-p", str(mem_p_local), "--batch-log2", str(batch_log2)])
    bench.late_imports()
    P = bench.Stream(args, 0, world, mem_p_agg).P
    for lo in range(1, P, 1 << 22):
        keys = np.arange(lo, min(lo + (1 << 22), P), dtype=np.uint32)
        O.insert(og, agg, W.bin_inserts(W.make_ielem(keys, None, W.KEYING_MIX), mem_p_agg, 3, W.BIN_REGION))
    hits = total = 0
    for step in range(steps):
        for r in range(world):
            gets = np.ascontiguousarray(data[r][f"gets{step}"])
            exp = O.search(og, agg, gets)
            assert np.array_equal(data[r][f"out{step}"], exp), f"rank {r} step {step}: search results differ"
            hits += int(((exp[:, 0] != 0) | (exp[:, 1] != 0)).sum())
            total += gets.shape[0]
        O.insert(og, agg, [np.ascontiguousarray(data[r][f"sets{step}"]) for r in range(world)])
    assert hits == total, "every GET key of this workload was pre-loaded"
    nb_local = 1 << (mem_p_local - 6)
    for r in range(world):
        assert np.array_equal(data[r]["table"], agg[r * nb_local:(r + 1) * nb_local]), f"shard {r} differs"
