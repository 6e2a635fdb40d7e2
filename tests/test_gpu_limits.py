"""Capacity of the update workspace at the boundary.  The reference accepts any count
(libgpuhash/gpu_hash.cu:521-556); here a call without a size hint is bounded by the
workspace (MKV_REQ_CAP / mkv_reserve) and the counts are DEVICE memory, so the library only
learns of an oversize batch after the fact: it must not go on silently -- and it must not kill
the host process either (a key-value server): the next update call reports it on stderr, the
drops are counted (mkv_dropped_batches), and work goes on."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, numpy as np, torch
import megakv_amd as M
from oracle import oracle as O
g = M.Geometry(mem_p=20)
idx = M.HashIndex(g)
rng = np.random.default_rng(5)
e = rng.integers(1, 2**32, size=(5000, 3), dtype=np.uint64).astype(np.uint32)
t = torch.from_numpy(e.view(np.int32)).cuda()
mode = sys.argv[1]
if mode == "hinted":
    idx.insert([t])                       # the host mirror passes the batch size: the workspace grows
    torch.cuda.synchronize()
    exp = O.new_table(O.geom(mem_p=20)); O.insert(O.geom(mem_p=20), exp, [e])
    assert np.array_equal(idx.table_numpy(), exp)
    print("hinted ok")
else:
    idx.insert([t], hint=0)               # the plain ABI call: no size known on the host
    torch.cuda.synchronize()
    print("first call returned", flush=True)
    assert idx.lib.mkv_dropped_batches() == 1, idx.lib.mkv_dropped_batches()
    small = t[:10].contiguous()
    idx.insert([small], hint=0)           # the next call on that workspace reports it -- and is applied
    torch.cuda.synchronize()
    exp = O.new_table(O.geom(mem_p=20)); O.insert(O.geom(mem_p=20), exp, [e[:10]])
    assert np.array_equal(idx.table_numpy(), exp), "the oversize batch was dropped whole, the next one applied"
    assert idx.lib.mkv_dropped_batches() == 1
    idx.insert([t], hint=0)               # a second oversize batch is counted too
    torch.cuda.synchronize()
    assert idx.lib.mkv_dropped_batches() == 2
    print("went on")
"""


def run(mode):
    env = dict(os.environ, MKV_REQ_CAP="2048", PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, "-c", CODE, mode], cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=600)


def test_hinted_insert_grows_the_workspace():
    p = run("hinted")
    assert p.returncode == 0 and "hinted ok" in p.stdout, p.stderr[-3000:]


def test_oversize_unhinted_batch_is_loud():
    p = run("plain")
    assert p.returncode == 0, p.stderr[-3000:]
    assert "first call returned" in p.stdout and "went on" in p.stdout
    assert "more requests than the update workspace holds" in p.stderr and "dropped" in p.stderr
