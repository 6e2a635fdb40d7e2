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


def run_and_check(world, mode, mem_p_local, scheme, seed, steps=4, timeout=600):
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
        # worker output goes to files: a full pipe on one rank would stall every rank at the next collective
        logs = [open(os.path.join(d, f"rank{r}.log"), "w") for r in range(world)]
        procs = [subprocess.Popen([sys.executable, "-m", "tests.sharded_worker", str(r), str(world), str(port),
                                   mode, d, str(mem_p_local), str(scheme), str(seed)], cwd=ROOT, env=env,
                                  stdout=logs[r], stderr=subprocess.STDOUT)
                 for r in range(world)]
        try:
            for p in procs:
                p.wait(timeout=timeout)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for f in logs:
                f.close()
        for r, p in enumerate(procs):
            tail = open(os.path.join(d, f"rank{r}.log")).read()[-3000:]
            assert p.returncode == 0, f"rank {r} failed:\n{tail}"
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
