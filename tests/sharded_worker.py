"""One rank of a sharded-index test run (launched by tests/test_sharded*.py).

  python -m tests.sharded_worker <rank> <world> <port> <mode> <outdir> <mem_p_local> <scheme> <seed>

mode "cpu": orchestration on CPU tensors, gloo collective, device operations
replaced by numpy + the CPU oracle (test infrastructure) -- exercises
ShardedIndex's geometry, slot format, canonical order and result routing.
mode "gpu": the HIP library on cuda:0 (all ranks share the one GPU of the box),
gloo collective staged through host memory.
Each rank writes its inputs, search results and final local table to outdir.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from megakv_amd.sharded import ShardedIndex, StagedComm, shard_geometry  # noqa: E402


class GlooComm:
    def all_to_all(self, out, inp):
        dist.all_to_all_single(out, inp)


class CpuOps:
    """numpy/oracle stand-in for HipOps (tests only)."""

    def __init__(self, geom):
        from oracle import oracle as O
        self.O = O
        self.og = O.geom(mem_p=geom.mem_p, scheme=geom.scheme, iblock_p=geom.iblock_p, max_cuckoo=geom.max_cuckoo)
        self.table = O.new_table(self.og)
        self._recv = {}

    def slot_words(self, cap_get, cap_set):
        return 4 + 2 * cap_get + 3 * cap_set

    def scratch_bytes(self, n_get, n_set, g):
        return 64

    def slot_set_pointers(self, recvbuf, g, slot_words, cap_get):
        ptrs = torch.zeros(g, dtype=torch.int64)
        self._recv[ptrs.data_ptr()] = (recvbuf, slot_words, cap_get)     # one entry per lane
        return ptrs

    def pack(self, gets, sets, agg_mask, shift, g, sendbuf, cap_get, cap_set, perm, scratch, overflow, stream=None):
        sb = sendbuf.numpy().view(np.uint32)
        pm = perm.numpy()
        G = gets.numpy().view(np.uint32)
        S = sets.numpy().view(np.uint32).reshape(-1, 3)
        sb[:, :4] = 0
        for s in range(g):
            ig = np.nonzero(((G[:, 1] & agg_mask) >> shift) == s)[0]
            isx = np.nonzero(((S[:, 1] & agg_mask) >> shift) == s)[0]
            if len(ig) > cap_get or len(isx) > cap_set:
                overflow[0] = 1
                continue
            sb[s, 0], sb[s, 1] = len(ig), len(isx)
            sb[s, 4:4 + 2 * len(ig)] = G[ig].reshape(-1)
            pm[s * cap_get:s * cap_get + len(ig)] = ig
            off = 4 + 2 * cap_get
            sb[s, off:off + 3 * len(isx)] = S[isx].reshape(-1)

    def search_slots(self, recvbuf, g, cap_get, cap_set, out_slots, set_counts, stream=None):
        rb = recvbuf.numpy().view(np.uint32)
        osl = out_slots.numpy().view(np.uint32)
        for s in range(g):
            n = int(rb[s, 0])
            q = np.ascontiguousarray(rb[s, 4:4 + 2 * n].reshape(n, 2))
            res = np.zeros((n, 2), np.uint32)
            self.O.search(self.og, self.table, q, out=res)
            osl[s, :n] = res
            set_counts[s] = int(rb[s, 1])

    def insert_slots(self, ptrs, set_counts, g, hint=0, stream=None):
        recvbuf, slot_words, cap_get = self._recv[ptrs.data_ptr()]
        rb = recvbuf.numpy().view(np.uint32)
        off = 4 + 2 * cap_get
        blocks = [np.ascontiguousarray(rb[s, off:off + 3 * int(set_counts[s])].reshape(-1, 3)) for s in range(g)]
        self.O.insert(self.og, self.table, blocks)

    def unpack(self, back, perm, sendbuf, g, cap_get, cap_set, out, stream=None):
        bk = back.numpy().view(np.uint32)
        sb = sendbuf.numpy().view(np.uint32)
        pm = perm.numpy()
        o = out.numpy().view(np.uint32)
        for s in range(g):
            n = int(sb[s, 0])
            o[pm[s * cap_get:s * cap_get + n]] = bk[s, :n]


def make_requests(rng, n_get, n_set, sig_space, known):
    gets = np.empty((n_get, 2), np.uint32)
    gets[:, 0] = rng.integers(1, sig_space + 1, size=n_get)
    gets[:, 1] = rng.integers(0, 2**32, size=n_get, dtype=np.uint64)
    if len(known):
        pick = known[rng.integers(0, len(known), size=n_get // 2)]
        gets[: n_get // 2] = pick[:, :2]
    sets = np.empty((n_set, 3), np.uint32)
    sets[:, 0] = rng.integers(1, sig_space + 1, size=n_set)
    sets[:, 1] = rng.integers(0, 2**32, size=n_set, dtype=np.uint64)
    sets[:, 2] = rng.integers(1, 2**32, size=n_set, dtype=np.uint64)
    return gets, sets


def zipf_main(rank, world, outdir, mem_p_local, scheme):
    """BASELINE configs[4] shape on one GPU: bench.py's Stream of this rank (95/5 GET/SET,
    Zipf-0.99, K-mix, fresh SET ids of this rank's range) through bench.py's pre-load and the
    two-lane pipeline; exchange staged through host memory (all ranks share cuda:0)."""
    import bench
    bench.late_imports()
    torch.cuda.set_device(0)
    k = world.bit_length() - 1
    batch_log2, steps = int(os.environ["MKV_ZIPF_BATCH_LOG2"]), int(os.environ["MKV_ZIPF_STEPS"])
    args = bench.parse(["--workload", "zipf95", "--mem-p", str(mem_p_local), "--batch-log2", str(batch_log2),
                        "--scheme", "cuckoo" if scheme == 0 else "2choice"])
    sg = bench.Stream(args, rank, world, mem_p_local + k)
    # a hot key sends all its GETs to one shard: slots twice the average share
    sh = bench.sharded_engine(mem_p_local, world, rank, scheme, "cuda:0", sg.n_get, sg.n_set, StagedComm(dist),
                              cap_factor=2.5)
    sh.preload(sg.P, sg.keying)
    rec, staged = {}, []
    for step in range(steps):
        gets, sets = sg.batch()
        rec[f"gets{step}"], rec[f"sets{step}"] = gets, sets
        staged.append(sh.stage(gets, sets))
    sh.run_pipelined(staged)
    torch.cuda.synchronize()
    for step in range(steps):
        rec[f"out{step}"] = staged[step]["out"].cpu().numpy().view(np.uint32)
    dist.barrier()
    assert int(sh.overflow[0]) == 0, "a slot overflowed"
    rec["table"] = sh.idx.table_numpy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **rec)
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    mode, outdir = sys.argv[4], sys.argv[5]
    mem_p_local, scheme, seed = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    if mode == "gpu-zipf":
        return zipf_main(rank, world, outdir, mem_p_local, scheme)
    n_get, n_set, steps = 3000, 1200, 4
    geom = shard_geometry(mem_p_local, world, scheme)
    if mode == "cpu":
        ops = CpuOps(geom)
        sh = ShardedIndex(mem_p_local, world, rank, scheme, "cpu", n_get, n_set, GlooComm(), ops=ops, cap_factor=2.0)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32))
    else:
        torch.cuda.set_device(0)
        sh = ShardedIndex(mem_p_local, world, rank, scheme, "cuda:0", n_get, n_set, StagedComm(dist), cap_factor=2.0)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
    rng = np.random.default_rng(seed * 100 + rank)
    known = np.zeros((0, 3), np.uint32)
    rec = {}
    if mode == "gpu-pipe":
        # all steps through the two-lane software pipeline
        staged = []
        for step in range(steps):
            gets, sets = make_requests(rng, n_get, n_set, 1 << 22, known)
            known = np.concatenate([known, sets])
            rec[f"gets{step}"], rec[f"sets{step}"] = gets, sets
            staged.append({"q": to_t(gets), "s": to_t(sets),
                           "out": torch.zeros((n_get, 2), dtype=torch.int32, device=sh.device)})
        sh.run_pipelined(staged)
        torch.cuda.synchronize()
        for step in range(steps):
            rec[f"out{step}"] = staged[step]["out"].cpu().numpy().view(np.uint32)
        dist.barrier()
        steps = 0
    for step in range(steps):
        gets, sets = make_requests(rng, n_get, n_set, 1 << 22, known)
        known = np.concatenate([known, sets])
        out = torch.zeros((n_get, 2), dtype=torch.int32, device=sh.device)
        sh.step_tensors(to_t(gets), to_t(sets), out)
        if mode == "gpu":
            torch.cuda.synchronize()
        rec[f"gets{step}"], rec[f"sets{step}"] = gets, sets
        rec[f"out{step}"] = out.cpu().numpy().view(np.uint32)
        dist.barrier()
    assert int(sh.overflow[0]) == 0
    rec["table"] = ops.table if mode == "cpu" else sh.idx.table_numpy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **rec)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
