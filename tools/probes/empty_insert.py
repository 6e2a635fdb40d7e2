#!/usr/bin/env python3
"""Launch cost of the insert kernels with next to nothing to do: 3000 inserts of 16 requests.
usage: empty_insert.py [path to an alternative libmegakv_hip.so]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import megakv_amd.lib as ML
if len(sys.argv) > 1:
    ML.LIB_PATH = sys.argv[1]
import megakv_amd as M
g = M.Geometry(mem_p=30)
idx = M.HashIndex(g)
rng = np.random.default_rng(1)
e = rng.integers(1, 2**32, size=(16, 3), dtype=np.uint64).astype(np.uint32)
t = torch.from_numpy(e.view(np.int32)).cuda()
ptrs, cnts, keep = idx.stage_blocks([t])
idx.insert_staged(ptrs, cnts, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3000):
    idx.insert_staged(ptrs, cnts, 1)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3000 * 1e6:.2f} us per insert call")
