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


DROPIN = r"""
import ctypes, numpy as np, torch
import megakv_amd as M
from megakv_amd.lib import load_library
from oracle import oracle as O
L = load_library()
g = M.Geometry(mem_p=30)                   # the drop-in entry points have their geometry baked in
og = O.geom(mem_p=30)
table = torch.zeros(g.ht_size // 4, dtype=torch.int32, device="cuda")
rng = np.random.default_rng(6)
n = 40000                                  # twenty times what the workspace is told to hold (MKV_REQ_CAP=2048)
e = np.empty((n, 3), np.uint32)
e[:, 0] = rng.integers(1, 2**32, size=n, dtype=np.uint64)
e[:, 1] = rng.integers(0, 2**32, size=n, dtype=np.uint64)
e[:, 2] = rng.integers(1, 2**32, size=n, dtype=np.uint64)
region = (e[:, 1] & g.hash_mask) >> (30 - 6 - 3)
blocks = [np.ascontiguousarray(e[region == b]) for b in range(8)]
dblocks = [torch.from_numpy(b.view(np.int32)).cuda() for b in blocks]
ptrs = torch.tensor([b.data_ptr() for b in dblocks], dtype=torch.int64).cuda()
counts_d = torch.zeros(8, dtype=torch.int32, device="cuda")
counts_h = np.array([b.shape[0] for b in blocks], dtype=np.int32)       # host memory, pageable
torch.cuda.synchronize()
# the reference's way: the counts go to the device with the library's own copy (src/mega_scheduler.c:487-494)
L.cudaMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
assert L.cudaMemcpyAsync(counts_d.data_ptr(), counts_h.ctypes.data, counts_h.nbytes, 1, None) == 0
M.gpu_hash_insert(table, ptrs, counts_d, 8, None)
torch.cuda.synchronize()
exp = O.new_table(og)
O.insert(og, exp, blocks)
got = table.cpu().numpy().view(np.uint32).reshape(-1, 16)
touched = np.unique(np.nonzero(exp.any(axis=1))[0])
assert np.array_equal(got[touched], exp[touched]) and int(np.count_nonzero(got)) == int(np.count_nonzero(exp))
assert L.mkv_dropped_batches() == 0
print("sized by the counts the host copied")
"""


def test_oversize_batch_through_the_dropin_names_is_run_in_passes():
    """gpu_hash_insert has no size argument (ref libgpuhash/gpu_hash.cu:521-556 takes any count): a host that copies
    its block counts to the device through the library's cudaMemcpyAsync -- the reference's scheduler does,
    src/mega_scheduler.c:487-494 -- has shown them to the library, which sizes the batch by them: forty thousand
    requests against a workspace told to hold two thousand are applied, exactly, nothing dropped."""
    env = dict(os.environ, MKV_REQ_CAP="2048", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", DROPIN], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sized by the counts the host copied" in p.stdout, p.stderr[-3000:]
