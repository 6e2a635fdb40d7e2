import os, sys, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29521")
os.dup2(2, 1)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
a = torch.zeros((1, 2_200_000), dtype=torch.int32, device="cuda"); b = torch.zeros_like(a)
pg = dist.group.WORLD
for name, fn in (("all_to_all_single", lambda: dist.all_to_all_single(b, a)),
                 ("pg.alltoall_base", lambda: pg.alltoall_base(b, a, [], []).wait()),
                 ("event record+wait", lambda: (e.record(), s2.wait_event(e))),
                 ("set_stream x2", lambda: (torch.cuda.set_stream(s2), torch.cuda.set_stream(s1)))):
    e = torch.cuda.Event(); s1 = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    sys.stderr.write(f"{name}: host {1e6*(t1-t0)/300:.1f} us/call, total {1e6*(t2-t0)/300:.1f} us/call\n")
dist.destroy_process_group()
