"""The worker cycle as a three-stage pipeline over pinned HOST buffers (mkv_pipe_*; reference:
src/mega_scheduler.c:393-504 with the triple buffering of src/include/mega_batch.h:74-97):
copy-in, cycle and copy-out of neighbouring batches overlap, results and table are those of
the cycles run one after the other -- checked against the oracle, pairs and resolved form."""
import ctypes

import numpy as np
import pytest
import torch

import megakv_amd as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def pinned(L, keep, arr):
    ptr = L.mkv_host_alloc(max(arr.nbytes, 64))
    assert ptr
    keep.append(ptr)
    buf = (ctypes.c_char * max(arr.nbytes, 64)).from_address(ptr)
    h = np.frombuffer(buf, dtype=arr.dtype, count=arr.size).reshape(arr.shape)
    h[...] = arr
    return h, ptr


@pytest.mark.parametrize("scheme,resolve,depth", [(0, 0, 3), (0, 1, 3), (1, 0, 2), (0, 0, 5)])
def test_pipeline_equals_cycles_one_after_the_other(scheme, resolve, depth):
    g = M.Geometry(mem_p=20, scheme=scheme)
    og = O.geom(mem_p=20, scheme=scheme)
    idx = M.HashIndex(g)
    L = idx.lib
    exp = O.new_table(og)
    rng = np.random.default_rng(40 + scheme + depth)
    keep, batches = [], []
    pool = rng.integers(1, 2**32, size=(60000, 2), dtype=np.uint64).astype(np.uint32)
    n_get_max, n_set_max, nblk = 30000, 9000, 8
    for i in range(11):
        ng = n_get_max if i % 4 != 2 else int(rng.integers(0, n_get_max))
        ns = n_set_max if i % 5 != 3 else int(rng.integers(0, 50))
        gets = pool[rng.integers(0, len(pool), size=ng)].copy()
        se = np.empty((ns, 3), np.uint32)
        se[:, :2] = pool[rng.integers(0, len(pool), size=ns)]
        se[:, 2] = rng.integers(1, 2**32, size=ns, dtype=np.uint64)
        region = ((se[:, 1] & g.hash_mask) >> (g.mem_p - 9)).astype(np.int64)
        blocks = [np.ascontiguousarray(se[region == b]) for b in range(nblk)]
        sets_all = np.concatenate(blocks) if ns else np.zeros((0, 3), np.uint32)
        hq, pq = pinned(L, keep, gets if ng else np.zeros((1, 2), np.uint32))
        hs, ps = pinned(L, keep, sets_all if ns else np.zeros((1, 3), np.uint32))
        ho, po = pinned(L, keep, np.full((max(ng, 1), 1 if resolve else 2), 0xEEEEEEEE, np.uint32))
        cnts = np.array([b.shape[0] for b in blocks], np.int32)
        want = O.search(og, exp, gets) if ng else np.zeros((0, 2), np.uint32)
        O.insert(og, exp, blocks)
        batches.append(dict(ng=ng, pq=pq, ps=ps, po=po, ho=ho, cnts=cnts, want=want))
    pipe = ctypes.c_void_p(0)
    assert L.mkv_pipe_create(ctypes.byref(pipe), ctypes.byref(idx._g), ctypes.c_void_p(idx.table.data_ptr()),
                             n_get_max, n_set_max, nblk, depth) == 0, L.mkv_pipe_error(pipe)
    try:
        m = len(batches)
        P, I64 = ctypes.c_void_p, ctypes.c_int64
        rc = L.mkv_pipe_run(pipe, m, (P * m)(*[b["pq"] for b in batches]), (I64 * m)(*[b["ng"] for b in batches]),
                            (P * m)(*[b["ps"] for b in batches]), (P * m)(*[b["cnts"].ctypes.data for b in batches]),
                            (P * m)(*[b["po"] for b in batches]), resolve,
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.mkv_pipe_error(pipe)
        torch.cuda.synchronize()
        for i, b in enumerate(batches):
            got = b["ho"][:b["ng"]]
            want = b["want"]
            if resolve:
                want = np.where(want[:, 0] != 0, want[:, 0], want[:, 1]).reshape(-1, 1)
            assert np.array_equal(got, want), f"batch {i}: results differ"
        assert np.array_equal(idx.table_numpy(), exp)
    finally:
        L.mkv_pipe_destroy(pipe)
        for ptr in keep:
            L.mkv_host_free(ptr)
