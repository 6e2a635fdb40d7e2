import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import megakv_amd as M
from megakv_amd import workload as W
from megakv_amd.sharded import ShardedIndex, NcclComm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.dup2(2, 1)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
n_get, n_set = 996148, 52428
sh = ShardedIndex(30, 1, 0, 0, "cuda:0", n_get, n_set, NcclComm(dist))
rng = np.random.default_rng(1)
def mk():
    g = rng.integers(1, 2**32, size=(n_get, 2), dtype=np.uint64).astype(np.uint32)
    s = rng.integers(1, 2**32, size=(n_set, 3), dtype=np.uint64).astype(np.uint32)
    return sh.stage(g, s)
bs = [mk() for _ in range(40)]
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    sh.reserve(st)
    for b in bs[:5]:
        sh.step(b, st, None)
    torch.cuda.synchronize()
    for name, fn in (("seq", lambda: [sh.step(b, st, None) for b in bs[5:]]), ("pipe", lambda: sh.run_pipelined(bs[5:]))):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        sys.stderr.write(f"{name}: host {1e6*(t1-t0)/35:.1f} us/step total {1e6*(t2-t0)/35:.1f} us/step\n")
    pr = cProfile.Profile(); pr.enable()
    for b in bs[5:]:
        sh.step(b, st, None)
    pr.disable(); torch.cuda.synchronize()
    import io
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); sys.stderr.write(s.getvalue()[:3500])
dist.destroy_process_group()
