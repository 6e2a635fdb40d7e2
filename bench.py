#!/usr/bin/env python3
"""bench.py -- Mops of the batched GPU hash index on MI355X.

One "step" = one pass of the hot path over one batch of 2^20 synthetic requests
already resident in HBM, driven the way Mega-KV's scheduler drives it
(reference src/mega_scheduler.c:393-504): clear the result buffer, search the
GETs, insert the SETs -- through the drop-in entry points' run-time-geometry
form (mkv_search / mkv_insert; gpu_hash_search / gpu_hash_insert are thin calls
of them, include/libgpuhash.h).

Default workload = BASELINE.json configs[2] (the one `metric` is quoted on):
MEM_P=30 (2^30-byte table), HASH_CUCKOO, batch 2^20, 95 % GET / 5 % SET,
Zipf-0.99 GET keys, table pre-loaded to load factor 0.2 through the insert
kernel, K-mix keying; `--workload get100` is configs[1].

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, aggregate table MEM_P = 30 + log2 N sharded by the
top bits of the bucket index, each rank feeds its own 2^20-request batches,
requests travel to their shard and results back with RCCL all-to-all
(megakv_amd/sharded.py).  Weak scaling.

Prints ONE JSON line (rank 0).  The oracle is used only for `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch and the HIP library are imported in main(), AFTER the decision whether this process is
# a launcher (--gpus N > 1 without a torch.distributed environment): a launcher never loads
# anything that could initialise the GPU.
torch = None
M = None
W = None

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GET_BYTES = 144                 # 8 (request) + 2 x 64 (buckets) + 8 (result)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="zipf95", choices=["zipf95", "get100", "zipf50"])
    ap.add_argument("--keying", default="mix", choices=["mix", "ref"])
    ap.add_argument("--sets", default="new", choices=["new", "upd"])
    ap.add_argument("--scheme", default="cuckoo", choices=["cuckoo", "2choice"])
    ap.add_argument("--sorted-gets", action="store_true",
                    help="experiment (DESIGN 5.1): the GETs of every batch pre-sorted by primary bucket on the host -- "
                         "the upper bound of what bucket-ordered probing could gain; the kernels are unchanged")
    ap.add_argument("--lf", type=float, default=0.2)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--mem-p", type=int, default=30, help="per-GPU table size (log2 bytes)")
    ap.add_argument("--memset", action="store_true",
                    help="clear the result buffer with a separate memset (reference call sequence) instead of "
                         "letting the search kernel write zeros on a miss")
    ap.add_argument("--overlap", default="fused", choices=["fused", "side", "none"],
                    help="where the insert's table-independent mark pass runs: inside the search launch, on a "
                         "side stream, or as its own kernel after the search")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded engine (pack / RCCL all-to-all / slots) even with one rank: a self-test "
                         "of the multi-GPU code path on a single GPU")
    ap.add_argument("--comm", default="nccl", choices=["nccl", "staged"],
                    help="staged: gloo collectives through host memory, all ranks on cuda:0 -- only to exercise "
                         "the multi-rank flow of this script on a one-GPU box")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region (exactly --steps steps between barriers) is run this many times, each time "
                         "from the same table image and over the same batches; `value` is the median")
    ap.add_argument("--no-abi", action="store_true",
                    help="skip the side measurement of the reference's three-call sequence (abi_three_call)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the C shard engine cannot be set up, go on with the Python-driven step over "
                         "torch.distributed instead of giving up (the default with several GPUs is to give up: a "
                         "fall-back line on an 8-GPU node is a wasted lease)")
    ap.add_argument("--preflight", action="store_true", help=argparse.SUPPRESS)     # (the child of preflight_check)
    ap.add_argument("--no-preflight", action="store_true",
                    help="skip the transport check a sharded run makes in a child process before it builds its table")
    ap.add_argument("--strict-engine", action="store_true",
                    help="N > 1: exit non-zero if the C shard engine cannot be set up or fails its warm-up check "
                         "(default: report the phase that failed, fall back to the Python-driven step and tag the "
                         'line "engine": "python-fallback")')
    ap.add_argument("--no-native", action="store_true",
                    help="sharded path: drive the step from Python/torch.distributed instead of the C engine")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N > 1: run the steps strictly one after the other (no exchange/compute overlap)")
    ap.add_argument("--events-every", type=int, default=0,
                    help="also time the search launch of every n-th step INSIDE the timed region (0 = never: events "
                         "cost a few microseconds of gaps per step; the roofline comes from the pass after it)")
    ap.add_argument("--kernel-launches", type=int, default=96,
                    help="launches of the dominant kernel timed by HIP events, one per step, in a separate pass "
                         "AFTER the timed region over the same batches (`value` stays unperturbed)")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed read-only searches before the warm-up steps, so that a short run (the driver's "
                         "--steps 20 --warmup 5 is 1.5 ms of GPU work) does not measure the clocks ramping up")
    ap.add_argument("--no-events", action="store_true", help="skip per-kernel HIP events (roofline = null)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed counter passes (profiles/search_traffic.json) instead of two "
                         "child passes of this command's workload under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; ~15 s each)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) side measurement")
    ap.add_argument("--no-sched", action="store_true", help="skip the scheduler-regime side measurement (16 workers' small cycles)")
    ap.add_argument("--cpu-batches", type=int, default=400,
                    help="batches of the same workload replayed on the CPU oracle (~10 s single-threaded)")
    ap.add_argument("--graph", action="store_true", help="replay each step from a captured hipGraph")
    ap.add_argument("--update-info", action="store_true",
                    help="diagnostics on stderr: the library's statistics of the last insert of the timed region")
    return ap.parse_args(argv)


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(device, non_blocking=False)


class Stream:
    """GET/SET key streams of SURVEY 8(d) for one rank."""

    def __init__(self, args, rank, world, mem_p_agg):
        self.args = args
        self.keying = W.KEYING_MIX if args.keying == "mix" else W.KEYING_REF
        self.P = W.preload_count(mem_p_agg, args.lf)            # keys 1 .. P-1 are pre-loaded
        self.mem_p_agg = mem_p_agg
        theta = 0.0 if args.workload == "get100" else 0.99
        # reference: mehcached_zipf_init(n = P-2, theta, seed 21); ranks differ by seed only
        self.zipf = W.ZipfKeys(self.P - 2, theta, 21 + rank)
        self.zipf_set = W.ZipfKeys(self.P - 2, theta, 1021 + rank)
        n = 1 << args.batch_log2
        self.n = n
        frac_set = {"zipf95": 0.05, "get100": 0.0, "zipf50": 0.5}[args.workload]
        self.n_set = int(n * frac_set)
        self.n_get = n - self.n_set
        # S-new: fresh sequential ids from P, a disjoint range per rank (src/mega_recv.c:671)
        span = ((1 << 31) - 1 - self.P) // max(world, 1)
        self.next_set = self.P + rank * span

    def batch(self):
        self.last_get_keys = self.zipf.keys(self.n_get)        # (kept for the result check after the warm-up)
        gets = W.make_selem(self.last_get_keys, self.keying)
        if getattr(self.args, "sorted_gets", False):
            order = np.argsort(gets[:, 1] & np.uint32((1 << (self.mem_p_agg - 6)) - 1), kind="stable")
            gets = np.ascontiguousarray(gets[order])
            self.last_get_keys = self.last_get_keys[order]
        if self.n_set == 0:
            return gets, np.zeros((0, 3), np.uint32)
        if self.args.sets == "new":
            k = np.arange(self.next_set, self.next_set + self.n_set, dtype=np.uint64).astype(np.uint32)
            self.next_set += self.n_set
        else:
            k = self.zipf_set.keys(self.n_set)
        return gets, W.make_ielem(k, None, self.keying)


def late_imports():
    """torch, the HIP library's host mirror and the workload generator (tests that drive
    Stream / SingleGpu directly call this first)."""
    global torch, M, W
    import torch as _torch
    import megakv_amd as _M
    import workload as _W
    torch, M, W = _torch, _M, _W


def launch_ranks(n):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as a
    CHILD process (one rank per GPU, rendezvous on 127.0.0.1), relay its output -- rank 0's one
    JSON line on stdout -- and return its exit code.  This process has not touched the GPU and
    never does; nothing is exec'ed."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


PREFLIGHT_S = 60.0              # per phase of the transport check
PREFLIGHT_STARTUP_S = 300.0     # ... and for its child process to come up (a first `import torch` on a fresh box)


def preflight_child(args):
    """The transport of a sharded run, alone, on a table of 1 MiB: the process group over RCCL, a 64-KiB
    send / receive ring over torch.distributed, then the C shard engine's own communicator -- create, connect
    (its one-word all-to-all and grouped send / receive with every peer) -- and two batches through
    mkv_shard_run against what a single table answers.  Every phase is announced on stderr before it starts:
    the parent reports the last one if this process dies or does not come back."""
    def phase(name):
        print(f"preflight phase: {name}", file=sys.stderr, flush=True)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    late_imports()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    phase("process group (RCCL)")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if world > 1:
        phase("64-KiB send / receive ring (torch.distributed over RCCL)")
        a = torch.full((16384,), rank, dtype=torch.int32, device=device)
        b = torch.empty_like(a)
        ops = [dist.P2POp(dist.isend, a, (rank + 1) % world), dist.P2POp(dist.irecv, b, (rank - 1) % world)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        torch.cuda.synchronize()
        assert int(b[0].item()) == (rank - 1) % world and bool((b == b[0]).all().item()), "ring: wrong payload"
    phase("C shard engine: create / connect (own RCCL communicator, first exchange)")
    scheme = M.SCHEME_CUCKOO if args.scheme == "cuckoo" else M.SCHEME_2CHOICE
    from megakv_amd.sharded import NcclComm
    n_get, n_set = 4096, 512
    eng = sharded_engine(20, world, rank, scheme, device, n_get, n_set, NcclComm(dist), cap_factor=2.0 * max(1, world // 2))
    eng.enable_native(dist)
    phase("C shard engine: two batches through mkv_shard_run")
    rng = np.random.default_rng(77 + rank)
    keys = rng.integers(1, 1 << 30, size=n_set, dtype=np.uint64).astype(np.uint32)
    sets = W.make_ielem(keys, None, W.KEYING_MIX)
    gets = W.make_selem(keys[rng.integers(0, n_set, size=n_get)], W.KEYING_MIX)
    b0 = eng.stage(np.zeros((0, 2), np.uint32), sets)           # this rank's keys in ...
    b1 = eng.stage(gets, np.zeros((0, 3), np.uint32))           # ... and looked up again, wherever their shard is
    eng.run_pipelined([b0, b1])
    torch.cuda.synchronize()
    assert not eng.overflowed(), "a slot overflowed"
    o = b1["out"].cpu().numpy().view(np.uint32)
    found = np.where(o[:, 0] != 0, o[:, 0], o[:, 1])
    want = gets[:, 0] * 0
    look = {int(k): int(l) for k, l in zip(sets[:, 0].tolist(), sets[:, 2].tolist())}
    want = np.array([look[int(sg)] for sg in gets[:, 0].tolist()], dtype=np.uint32)
    assert np.array_equal(found, want), f"{int((found != want).sum())} of {n_get} GETs came back wrong"
    eng.native.close()
    dist.destroy_process_group()
    print("preflight ok", file=sys.stderr, flush=True)
    return 0


def preflight_check(args):
    """Before a sharded run over RCCL touches the GPU or builds its table: the transport check above in a CHILD
    process of this rank (the children of all ranks meet on the next port) under a deadline.  Returns None, or
    the phase that failed or hung and what the child said last."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = env.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29517")) + 1)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # (rank 0's child serves the children's store itself)
    cmd = [sys.executable, os.path.abspath(__file__), "--preflight", "--scheme", args.scheme]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    # The deadline runs per PHASE, from the line that announces it: importing torch in a fresh container can take a
    # minute or two by itself (the image pages in) and is no sign of a transport that hangs.
    import threading
    seen = {"phases": [], "t": time.time(), "tail": []}

    def reader():
        for ln in child.stderr:
            ln = ln.rstrip()
            if ln.startswith("preflight phase:"):
                seen["phases"].append(ln.split(":", 1)[1].strip())
                seen["t"] = time.time()
            elif ln:
                seen["tail"] = (seen["tail"] + [ln])[-6:]
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    timed_out = False
    while child.poll() is None:
        limit = PREFLIGHT_S if seen["phases"] else PREFLIGHT_STARTUP_S
        if time.time() - seen["t"] > limit:
            child.kill()                             # (this very process: the PID we started)
            timed_out = True
            break
        time.sleep(0.2)
    child.wait()
    th.join(timeout=5)
    if child.returncode == 0 and not timed_out:
        return None
    last = seen["phases"][-1] if seen["phases"] else "start-up (importing torch)"
    tail = " | ".join(seen["tail"])
    return (f"{'no answer within %.0f s in' % (PREFLIGHT_S if seen['phases'] else PREFLIGHT_STARTUP_S) if timed_out else 'failed in'} "
            f"phase \"{last}\"" + (f": {tail}" if tail else ""))


def main():
    # Libraries (RCCL prints a version banner) must not share the channel of the one JSON
    # line: everything written to fd 1 from here on goes to stderr, the line is written
    # to the saved descriptor at the end.
    args = parse()
    if args.preflight:
        raise SystemExit(preflight_child(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.force_sharded) and args.comm == "nccl" and not args.no_native
            and not args.no_pipeline and not args.no_preflight):
        # (this process has not touched the GPU yet: the check runs in a child, nothing is exec'ed here)
        why = preflight_check(args)
        if why:
            print(f"[rank {os.environ.get('RANK', '0')}] bench.py: the transport check before the run {why}; giving up "
                  f"(--no-preflight skips the check)", file=sys.stderr, flush=True)
            raise SystemExit(23)
    # (before anything initialises HIP: the host driver of this pool only supports dmabuf IPC -- RCCL's
    # peer mappings fail with "hipIpcGetMemHandle: invalid argument" without it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    late_imports()
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if args.comm == "staged":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist_mod
        dist = dist_mod
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        if args.comm == "staged":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    log2w = world.bit_length() - 1
    assert 1 << log2w == world and world <= 8, "1/2/4/8 GPUs"
    mem_p_agg = args.mem_p + log2w
    scheme = M.SCHEME_CUCKOO if args.scheme == "cuckoo" else M.SCHEME_2CHOICE

    stream_gen = Stream(args, rank, world, mem_p_agg)
    # ---- distinct pre-generated batches (host side first: the sharded engine sizes its slots by them)
    nb = args.steps + args.warmup
    t0 = time.time()
    host_batches, check_keys = [], None
    for i in range(nb):
        host_batches.append(stream_gen.batch())
        if i == args.warmup - 1:
            check_keys = stream_gen.last_get_keys.copy()
    t_gen = time.time() - t0
    slot_factor = None
    if not sharded:
        from megakv_amd.index import HashIndex
        geom = M.Geometry(mem_p=args.mem_p, scheme=scheme)
        engine = SingleGpu(HashIndex(geom, device=device), args)
    else:
        from megakv_amd.sharded import NcclComm, StagedComm
        # Slot capacity: the largest share one shard gets of any batch of any rank (a hot key sends
        # all its GETs to one shard), plus 3 %.  Padding is exchanged like payload, so no more than
        # needed; and a slot can then not overflow in this run.
        mask, shift, share = (1 << (mem_p_agg - 6)) - 1, args.mem_p - 6, 1.0
        for gets, sets in host_batches:
            for a in (gets, sets):
                if a.shape[0] >= world:
                    c = np.bincount((a[:, 1] & np.uint32(mask)) >> np.uint32(shift), minlength=world)
                    share = max(share, float(c.max()) * world / a.shape[0])
        t = torch.tensor([share], dtype=torch.float64, device="cpu" if args.comm == "staged" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        slot_factor = round(float(t.item()) * 1.03, 4)
        engine = sharded_engine(args.mem_p, world, rank, scheme, device, stream_gen.n_get, stream_gen.n_set,
                              StagedComm(dist) if args.comm == "staged" else NcclComm(dist),
                              cap_factor=slot_factor)
    native = False
    engine_error = None               # why the C shard engine is not what ran (phase and message), if it is not
    strict = args.strict_engine or (world > 1 and not args.allow_fallback)
    if sharded and args.comm == "nccl" and not args.no_native and not args.no_pipeline:
        # the step as ONE library call per run (own RCCL communicator); all ranks agree, or all
        # fall back to the Python-driven step over torch.distributed
        ok = 1
        import threading
        wd = threading.Timer(300.0, lambda: (print("bench.py: RCCL communicator setup did not finish in 300 s",
                                                   file=sys.stderr, flush=True), os._exit(19)))
        wd.daemon = True
        wd.start()
        try:
            engine.enable_native(dist)
        except Exception as ex:                                   # noqa: BLE001
            engine_error = f"set-up: {ex}"
            print(f"[rank {rank}] C shard engine FAILED during set-up (create / connect / first exchange): {ex}",
                  file=sys.stderr, flush=True)
            ok = 0
        wd.cancel()
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if engine.native is not None:
                engine.native.close()
            engine.native = None
            engine_error = engine_error or "set-up: another rank failed (see its message)"
            if strict:
                print(f"[rank {rank}] giving up ({engine_error}); --allow-fallback goes on with the Python-driven step",
                      file=sys.stderr, flush=True)
                dist.destroy_process_group()
                os._exit(21)
            print(f"[rank {rank}] falling back to the Python-driven step over torch.distributed; the line will say "
                  f'"engine": "python-fallback"', file=sys.stderr, flush=True)
        native = engine.native is not None

    # ---- pre-load: keys 1 .. P-1, loc = key, through the insert kernel ----------
    # (on the stream the steps will run on: a host that updates its table from ONE stream never makes
    # the library order updates across streams)
    run_stream = torch.cuda.Stream(device=device)
    t0 = time.time()
    with torch.cuda.stream(run_stream):
        engine.preload(stream_gen.P, stream_gen.keying)
    torch.cuda.synchronize()
    t_preload = time.time() - t0

    # ---- the batches, resident in HBM --------------------------------------------------
    t0 = time.time()
    batches = [engine.stage(g, s) for g, s in host_batches]
    last_sets = [s_ for _, s_ in host_batches[-4:]] if args.steps >= 1 else []
    del host_batches
    torch.cuda.synchronize()
    t_gen += time.time() - t0

    use_events = not args.no_events
    ev = []

    bar_word = torch.zeros(1, dtype=torch.int32, device=device) if (dist is not None and args.comm != "staged") else None

    def barrier():
        """All ranks meet, then the device is idle.  Over RCCL the meeting is a one-word all-reduce
        enqueued on the CURRENT stream -- behind the work queued there, no host synchronisation before
        it -- followed by one device synchronisation: a rank leaves when every rank's work is done
        (dist.barrier() is the same collective behind a host synchronisation of its own: 160 us
        between the end of the work and the end of the region with one rank, 30 us this way)."""
        if dist is not None:
            if bar_word is not None:
                dist.all_reduce(bar_word)
            else:
                dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(run_stream):
        engine.reserve(run_stream)
        # a collective that never completes would hang the box: give the warm-up a deadline
        import threading
        watchdog = threading.Timer(300.0, lambda: (print("bench.py: warm-up did not finish in 300 s, giving up",
                                                         file=sys.stderr, flush=True), os._exit(19)))
        watchdog.daemon = True
        watchdog.start()
        t_prewarm = engine.prewarm(batches, args.prewarm_ms, run_stream) if args.prewarm_ms > 0 else 0.0
        if sharded and not args.no_pipeline:
            engine.run_pipelined(batches[:args.warmup])       # warms both lanes and their workspaces
        else:
            for b in batches[:args.warmup]:
                engine.step(b, run_stream, None)
        run_stream.synchronize()
        watchdog.cancel()
        if native and args.warmup and args.lf <= 0.5:
            # every GET key was pre-loaded or set earlier: anything but a full hit rate means the
            # exchange went wrong -> all ranks drop to the Python-driven step and warm up again
            o = batches[args.warmup - 1]["out"]
            res0 = torch.where(o[:, 0] != 0, o[:, 0], o[:, 1]).cpu().numpy().view(np.uint32)
            good = torch.tensor([1 if float((res0 == check_keys).mean()) > 0.999 else 0],
                                dtype=torch.int32, device=device)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if int(good.item()) == 0:
                engine_error = "warm-up check: GETs of pre-loaded keys did not all come back with their key (on some rank)"
                print(f"[rank {rank}] C shard engine FAILED its warm-up check (hit rate of pre-loaded keys "
                      f"{float((res0 == check_keys).mean()):.4f} on this rank)", file=sys.stderr, flush=True)
                if strict:
                    dist.destroy_process_group()
                    os._exit(22)
                print(f"[rank {rank}] falling back to the Python-driven step; the line will say "
                      f'"engine": "python-fallback"', file=sys.stderr, flush=True)
                engine.native.close()
                engine.native = None
                native = False
                engine.run_pipelined(batches[:args.warmup])
                run_stream.synchronize()
        hit_rate = engine.hit_rate(batches[args.warmup - 1]) if args.warmup else None
        # stronger than a hit: the pre-load stored loc = key id, so every GET of a pre-loaded key
        # must come back with exactly its own key id, in its own place of the result array
        # (a result routed or un-permuted wrongly would still be "a hit")
        loc_ok = None
        if args.warmup and check_keys is not None and args.lf <= 0.5:
            o = batches[args.warmup - 1]["out"]
            res = torch.where(o[:, 0] != 0, o[:, 0], o[:, 1]).cpu().numpy().view(np.uint32)
            loc_ok = round(float((res == check_keys).mean()), 6)
        every = max(1, args.events_every)
        timer = engine.make_timer((args.steps + every - 1) // every) \
            if (use_events and not args.graph and args.events_every > 0) else None
        timed = batches[args.warmup:]
        if args.prewarm_ms > 0:
            # (the result checks above left the GPU idle for some milliseconds)
            t_prewarm += engine.prewarm(batches, min(args.prewarm_ms, 40.0), run_stream)
        # the argument arrays of the run (device pointers and counts of the K batches) are marshalled
        # before the clock starts: 100 tensor attribute look-ups in Python are not a step's work
        run_timed = None
        if sharded and not args.no_pipeline:
            run_timed = engine.prepare_pipelined(timed, (lambda i: [timer.h, i // every] if (timer and i % every == 0) else None))
        # The timed region -- barrier, EXACTLY K steps, barrier -- is run `repeats` times, every time from the
        # SAME table image (a device copy of the table taken here, put back before each run: 1 GiB, < 1 ms,
        # outside the clock) and over the SAME K batches, so that the runs do the same work; `value` is the
        # median, the others are in the line.  One sample is not a measurement: the same command read 15.2
        # and 19.9 Gops minutes apart in round 3 (profiles/r03_final_lines.txt).
        image = engine.table_image()
        abi = None
        if (not sharded and not args.no_abi and not args.graph and args.steps and not args.memset
                and args.overlap == "fused"):
            # side measurement, same table image, same batches, same clock: the reference's own call sequence
            # (src/mega_scheduler.c:393-502: memset, gpu_hash_search, gpu_hash_insert as three calls)
            # When the geometry is the one baked into the library's drop-in entry points (MEM_P 30, cuckoo, like the
            # reference's header) the calls are the literal symbols an unchanged host links -- cudaMemsetAsync,
            # gpu_hash_search, gpu_hash_insert -- eager, and once more with the library's deferred mode
            # (mkv_set_defer: the calls are recorded and issued as one cycle by the flush a synchronisation makes).
            literal = args.mem_p == 30 and args.scheme == "cuckoo"

            def abi_run(step):
                barrier()
                t0 = time.perf_counter()
                for b in timed:
                    step(b, run_stream)
                barrier()
                dt = time.perf_counter() - t0
                engine.restore_table(image)
                return {"value": round((1 << args.batch_log2) * args.steps / dt / 1e6, 1), "unit": "Mops",
                        "us_per_step": round(dt / args.steps * 1e6, 2)}
            if literal:
                abi = abi_run(engine.step_abi_literal)
                abi["calls"] = ("cudaMemsetAsync(out) + gpu_hash_search + gpu_hash_insert per step: the drop-in symbols "
                                "themselves, eager; same batches and table image as `value`")
                engine.idx.lib.mkv_set_defer(1)
                try:
                    abi["deferred"] = abi_run(engine.step_abi_deferred)
                finally:
                    engine.idx.lib.mkv_set_defer(0)
                abi["deferred"]["calls"] = ("the same three calls with the library's deferred mode on (MKV_DEFER=1) and an mkv_flush() "
                                            "per step where the reference's scheduler synchronises: issued as one mkv_cycle")
            else:
                abi = abi_run(engine.step_three_call)
                abi["calls"] = ("hipMemsetAsync(out) + mkv_search(KEEP_MISS) + mkv_insert_hinted per step (what gpu_hash_search / "
                                "gpu_hash_insert call with their baked-in geometry), same batches and table image as `value`")
        elapsed_runs, t_submit = [], 0.0
        for rep in range(max(1, args.repeats)):
            if rep:
                engine.restore_table(image)
            barrier()
            t_start = time.perf_counter()
            if run_timed is not None:
                # several batches in flight: exchanges of one overlap the local kernels of the others
                run_timed()
            else:
                for i, b in enumerate(timed):
                    engine.step(b, run_stream, [timer.h, i // every] if (timer and i % every == 0) else None)
            t_submit = time.perf_counter() - t_start
            barrier()                     # (on the run stream: behind the K steps)
            e = time.perf_counter() - t_start
            if dist is not None:
                t = torch.tensor([e], dtype=torch.float64, device="cpu" if args.comm == "staged" else device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e = float(t.item())
            elapsed_runs.append(e)
        del image
    elapsed = float(np.median(elapsed_runs))

    n_ops = (1 << args.batch_log2) * args.steps * world
    value = n_ops / elapsed / 1e6
    repeats = {"runs": len(elapsed_runs), "mops": [round(n_ops / e / 1e6, 1) for e in elapsed_runs],
               "us_per_step": [round(e / args.steps * 1e6, 2) for e in elapsed_runs],
               "spread": round((max(elapsed_runs) - min(elapsed_runs)) / elapsed, 4)}
    if args.update_info and not sharded:
        print("update info:", engine.idx.last_update_info(run_stream, cycle_insert=True), file=sys.stderr)

    # the results of the LAST timed batch, checked like those of the warm-up: every GET of a
    # pre-loaded key must have come back with its own key id in its own place (all ranks)
    loc_ok_last = None
    if args.steps and args.lf <= 0.5:
        o = batches[-1]["out"]
        res = torch.where(o[:, 0] != 0, o[:, 0], o[:, 1]).cpu().numpy().view(np.uint32)
        loc_ok_last = float((res == stream_gen.last_get_keys).mean())
        if dist is not None:
            t = torch.tensor([loc_ok_last], dtype=torch.float64, device="cpu" if args.comm == "staged" else device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            loc_ok_last = float(t.item())
        loc_ok_last = round(loc_ok_last, 6)

    # ---- what the timed region did, proven after it ---------------------------------------
    # (a) every SET key of the last timed batches is searched: the inserts of the timed region
    #     are in the table, under their own key id (loc = key id);
    # (b) the occupied slots of the table are counted: pre-loaded + inserted keys.
    set_ok, entries = None, None
    try:
        if last_sets and stream_gen.n_set:
            set_ok = engine.verify_sets(last_sets, run_stream)
        n_ins = stream_gen.n_set * (args.steps + args.warmup) * world if args.sets == "new" else 0
        entries = {"counted": engine.count_entries(dist), "expected": (stream_gen.P - 1) + n_ins}
        # exact only while nothing can be displaced: a cuckoo table that stays below half full (an
        # eviction chain of maximum length drops its last victim, ref gpu_hash.cu:414-423; a
        # 2-choice insert without room replaces an entry, :197-209)
        final_lf = entries["expected"] / float(8 << (mem_p_agg - 6))
        exact = args.scheme == "cuckoo" and final_lf <= 0.5
        entries["ok"] = entries["counted"] == entries["expected"] if exact else None
        if not exact:
            entries["displaced"] = entries["expected"] - entries["counted"]
    except Exception as ex:                                       # noqa: BLE001
        entries = {"error": f"{type(ex).__name__}: {ex}"}
    if dist is not None and set_ok is not None:
        t = torch.tensor([set_ok], dtype=torch.float64, device="cpu" if args.comm == "staged" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        set_ok = float(t.item())

    # ---- the dominant kernel, timed launch by launch in a pass of its own -------------------
    # The same cycles over the timed region's batches once more (their SETs are then updates
    # with the same value: the table does not change); the search launch of EVERY step delivers
    # its start and end time into a pair of HIP events (hipExtLaunchKernelGGL on the run stream:
    # the dispatch's own timestamps, what rocprofv3 reports as the kernel's duration -- a pair of
    # hipEventRecord calls around the launch reads 2-3 us more, the launch latency).  Not part
    # of `value`.
    roofline = None
    inside = np.array(timer.elapsed_ms()) if timer else None
    if use_events and not args.graph and args.steps:
        L = max(8, args.kernel_launches) + 4       # (4 more: the first ones are dropped below)
        post = [batches[args.warmup + (i % args.steps)] for i in range(L)]
        ptimer = engine.make_timer(L)
        with torch.cuda.stream(run_stream):
            if sharded and not args.no_pipeline:
                engine.run_pipelined(post, (lambda i: [ptimer.h, i]))
            else:
                for i, b in enumerate(post):
                    engine.step(b, run_stream, [ptimer.h, i])
            run_stream.synchronize()
        barrier()
        ms = np.array(ptimer.elapsed_ms())[4:]          # (the first launches after the checks above start cold)
        med_s = float(np.median(ms)) * 1e-3
        algo_bytes = GET_BYTES * engine.search_requests_per_launch(stream_gen)
        achieved = algo_bytes / med_s / 1e9
        fused = not sharded and args.overlap == "fused" and not args.memset and stream_gen.n_set > 0
        traffic, traffic_src, traffic_lib = load_traffic(args, fused, sharded)
        committed = {"bytes": traffic, "source": traffic_src, "library": traffic_lib}
        live_why = None
        if rank == 0 and world == 1 and not sharded and not args.no_live_traffic:
            # (measured by this run itself, in two child passes under the profiler; the committed passes stay beside it)
            try:
                live, live_why = live_traffic(args, fused)
            except Exception as ex:                               # noqa: BLE001  (whatever the profiler does, the line is printed)
                live, live_why = None, f"{type(ex).__name__}: {ex}"
            if live is not None:
                traffic, traffic_src, traffic_lib = live, live_why, library_id(engine)
                live_why = None
        roofline = {"bound": "hbm",
                    "kernel": "mkv::k_search_mark (search work-groups + the insert's mark work-groups)" if fused
                    else ("mkv::k_search_slots_mark" if sharded else "mkv::k_search"),
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    # (the library build the counter passes were taken with -- the last word of its mkv_version() -- beside
                    # this run's: a profile older than the kernels shows as two different words)
                    "traffic_head": traffic_lib, "library": library_id(engine),
                    "traffic_committed": committed, "traffic_not_live_because": live_why,
                    "algorithmic_bytes_per_launch": algo_bytes,
                    "launch_us": round(med_s * 1e6, 2),
                    "launch_us_is": "median over launches_timed launches of the kernel's own duration: HIP events that "
                                    "receive the dispatch's start and end time (hipExtLaunchKernelGGL on the run stream)",
                    "launches_timed": int(len(ms)), "avg_launch_us": round(float(ms.mean()) * 1e3, 2),
                    "p10_us": round(float(np.percentile(ms, 10)) * 1e3, 2),
                    "p90_us": round(float(np.percentile(ms, 90)) * 1e3, 2),
                    "timed_in": "a pass after the timed region, same batches, every step"}
        if inside is not None and len(inside):
            roofline["inside_timed_region"] = {"launches_timed": int(len(inside)),
                                               "median_us": round(float(np.median(inside)) * 1e3, 2)}

    # The C shard engine's own device timestamps, per rank: mean microseconds of a batch's pack, exchange (one
    # grouped launch: results of batch i back, requests of batch i+B out), local phase and un-pack over eight
    # batches of one more run with events around every operation (they cost 3 - 5 us each: weights of the
    # phases, not a ledger of the timed region) -- what a SCALE record is read against BASELINE.md section 4 with.
    phases = None
    if sharded and native and args.steps >= 26:
        # (the traced pass is a collective of its own -- every rank runs its exchanges -- and the gather behind it is one
        # too: a rank that failed must not leave the others waiting in them.  What a rank raised is kept, the ranks agree
        # on success with an all-reduce that every rank reaches, and only then enter the gather.)
        mine, err = None, None
        try:
            with torch.cuda.stream(run_stream):
                mine = engine.native.phases_us(batches[args.warmup:args.warmup + args.steps], run_stream)
        except Exception as ex:                                   # noqa: BLE001
            err = f"{type(ex).__name__}: {ex}"
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float32, device=device)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 0.5:
            phases = {"error": err or "another rank failed its traced pass"}
        else:
            t = torch.tensor([mine["pack"], mine["exchange"], mine["local"], mine["unpack"]], dtype=torch.float32, device=device)
            allr = [torch.zeros_like(t) for _ in range(world)]
            if world > 1:
                dist.all_gather(allr, t)
            else:
                allr = [t]
            phases = [{k: round(float(v), 2) for k, v in zip(("pack", "exchange", "local", "unpack"), a.tolist())} for a in allr]

    # side measurements: whatever happens in them, the line with `value` is still printed
    pcie = None
    if rank == 0 and not sharded and not args.no_pcie:
        try:
            pcie = pcie_inclusive(engine, batches[args.warmup:args.warmup + 24], run_stream, args)
        except Exception as ex:                                   # noqa: BLE001
            pcie = {"error": f"{type(ex).__name__}: {ex}"}

    # the reference's own operating point: one scheduler cycle of 16 workers at its batch sizes
    sched = None
    if rank == 0 and not sharded and not args.no_sched:
        try:
            sched = scheduler_regime(engine, run_stream)
        except Exception as ex:                                   # noqa: BLE001
            sched = {"error": f"{type(ex).__name__}: {ex}"}

    cpu = None
    if rank == 0 and not sharded and not args.no_cpu:
        try:
            cpu = cpu_baseline(engine, stream_gen, args)
        except Exception as ex:                                   # noqa: BLE001
            cpu = {"error": f"{type(ex).__name__}: {ex}"}

    if sharded and engine.overflowed():
        raise SystemExit("a shard slot overflowed: raise cap_factor (results of this run are invalid)")
    if rank == 0:
        line = {
            "metric": "Mops (search+insert) on Zipf-0.99, 2^30-B table, 1/2/4/8 GPUs; HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "Mops", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": {"zipf95": "BASELINE configs[2]: 95/5 GET/SET Zipf-0.99",
                                    "get100": "BASELINE configs[1]: 100% GET uniform",
                                    "zipf50": "BASELINE configs[3]: 50/50 GET/SET Zipf-0.99"}[args.workload],
                       "mem_p_per_gpu": args.mem_p, "mem_p_aggregate": mem_p_agg, "scheme": args.scheme,
                       "batch": 1 << args.batch_log2, "gets": stream_gen.n_get, "sets": stream_gen.n_set,
                       "keying": "K-" + args.keying, "set_keys": "S-" + args.sets, "load_factor": args.lf,
                       "preloaded_keys": stream_gen.P - 1, "miss_zeroing": "memset" if args.memset else "by the search kernel", "mark_pass": args.overlap,
                       "slot_capacity_factor": slot_factor,
                       "sharding": "single table" if not sharded else
                       f"{world} shards by top bucket bits, "
                       + ("all-to-all staged through host memory (gloo), " if args.comm == "staged" else "RCCL all-to-all, ")
                       + ("steps one by one" if args.no_pipeline else "2 batches in flight"),
                       "launch": "hipGraph" if args.graph else
                       ("one C call per run (library-owned RCCL communicator)" if native else "eager")},
            "repeats": repeats, "abi_three_call": abi,
            "engine": ("single table (mkv_cycle)" if not sharded else "C shard engine (mkv_shard_run)" if native else
                       "python-fallback" if engine_error else "python-driven step"),
            "engine_error": engine_error, "phases_us_per_rank": phases,
            "get_hit_rate": hit_rate, "get_loc_correct": loc_ok, "get_loc_correct_last_timed": loc_ok_last,
            "set_loc_correct_last_timed": set_ok, "table_entries": entries,
            "table_entries_ok": entries.get("ok") if isinstance(entries, dict) else None,
            "step_algorithmic": step_algorithmic(stream_gen, cpu, elapsed / args.steps),
            "roofline": roofline, "cpu_baseline": cpu, "pcie_inclusive": pcie, "scheduler_regime": sched, "prewarm_s": round(t_prewarm, 3),
            "host_submit_ms_per_step": round(t_submit / args.steps * 1e3, 5),
            "setup_s": {"preload": round(t_preload, 2), "generate": round(t_gen, 2)},
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def sharded_engine(*a, **kw):
    """ShardedIndex (megakv_amd/sharded.py) plus what this benchmark drives around it: pre-load,
    staging of a batch, clock pre-warm and the checks of the timed region's work."""
    from megakv_amd.sharded import ShardedIndex

    class ShardedBench(ShardedIndex):
        def preload(self, P, keying):
            g = self.geom
            step = 1 << 22
            self.idx.reserve(2 * step // self.world + 4096)
            for lo in range(1, P, step):
                k = np.arange(lo, min(lo + step, P), dtype=np.uint64).astype(np.uint32)
                e = W.make_ielem(k, None, keying)
                mine = np.take(e, np.flatnonzero(self.shard_of(e[:, 1]) == self.rank), axis=0)
                blocks = W.bin_inserts(mine, g.mem_p, g.iblock_p, W.BIN_REGION)
                self.idx.insert([torch.from_numpy(np.ascontiguousarray(b).view(np.int32)).to(self.device)
                                 for b in blocks])
            torch.cuda.synchronize()

        def stage(self, gets, sets):
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(self.device)
            return {"q": dev(gets), "s": dev(sets.reshape(-1, 3)),
                    "out": torch.zeros((gets.shape[0], 2), dtype=torch.int32, device=self.device)}

        def table_image(self):
            torch.cuda.synchronize()
            return self.idx.table.clone()

        def restore_table(self, image):
            torch.cuda.synchronize()
            self.idx.table.copy_(image)
            torch.cuda.synchronize()

        def reserve(self, stream):
            self.idx.reserve(self.world * self.cap_set + 1024, stream)
            for ln in self.lanes:
                if ln.stream is not None:
                    self.idx.reserve(self.world * self.cap_set + 1024, ln.stream)

        def make_timer(self, slots):
            from megakv_amd.index import SearchTimer
            return SearchTimer(slots)

        def step(self, b, stream, pair):
            self._timing = pair            # [timer, slot] or None
            self.step_tensors(b["q"], b["s"], b["out"])
            self._timing = None

        def hit_rate(self, b):
            o = b["out"]
            return round(float(((o[:, 0] != 0) | (o[:, 1] != 0)).float().mean().item()), 6)

        def search_requests_per_launch(self, sg):
            return sg.n_get

        def prewarm(self, batches, ms, stream):
            """untimed read-only searches of the local table (clocks up before a short run)"""
            scratch = torch.empty_like(batches[0]["out"])
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < ms:
                for b in batches[:16]:
                    if b["q"].shape[0] == scratch.shape[0]:
                        self.idx.search(b["q"], out=scratch, stream=stream)
                stream.synchronize()
            return time.perf_counter() - t0

        def verify_sets(self, sets_list, stream):
            """fraction of this rank's SET requests (host arrays [n,3]) found under their own
            location through the sharded path -- minimum over the batches.  Collective."""
            staged = [self.stage(np.ascontiguousarray(s[:, :2]), np.zeros((0, 3), np.uint32)) for s in sets_list]
            self.run_pipelined(staged)
            torch.cuda.synchronize()
            worst = 1.0
            for s, b in zip(sets_list, staged):
                if not s.shape[0]:
                    continue
                o = b["out"]
                res = torch.where(o[:, 0] != 0, o[:, 0], o[:, 1]).cpu().numpy().view(np.uint32)
                worst = min(worst, float((res == s[:, 2]).mean()))
            return round(worst, 6)

        def count_entries(self, dist=None):
            """occupied slots of all shards (sum over the ranks)"""
            n = (self.idx.table.view(-1, 16)[:, :8] != 0).sum().to(torch.int64)
            if dist is not None and self.world > 1:
                t = n.reshape(1).cpu() if dist.get_backend() == "gloo" else n.reshape(1)
                dist.all_reduce(t)
                return int(t.item())
            return int(n.item())

    return ShardedBench(*a, **kw)


class SingleGpu:
    def __init__(self, index, args):
        self.idx = index
        self.args = args
        self.device = index.device
        self.graphs = {}

    def preload(self, P, keying):
        g = self.idx.geom
        step = 1 << 20
        self.idx.reserve(step + 8)
        for lo in range(1, P, step):
            k = np.arange(lo, min(lo + step, P), dtype=np.uint32)
            blocks = W.bin_inserts(W.make_ielem(k, None, keying), g.mem_p, g.iblock_p, W.BIN_REGION)
            self.idx.insert([dev(b, self.device) for b in blocks])
        torch.cuda.synchronize()

    def stage(self, gets, sets):
        g = self.idx.geom
        b = {"q": dev(gets, self.device),
             "out": torch.empty((gets.shape[0], 2), dtype=torch.int32, device=self.device)}
        if sets.shape[0]:
            # the insert blocks of a batch live in ONE device buffer (block order), each block a slice
            parts = W.bin_inserts(sets, g.mem_p, g.iblock_p, W.BIN_REGION)
            b["sets_all"] = dev(np.concatenate(parts), self.device)
            offs = np.concatenate([[0], np.cumsum([p.shape[0] for p in parts])])
            blocks = [b["sets_all"][offs[i]:offs[i + 1]] for i in range(len(parts))]
            b["ptrs"], b["cnts"], b["keep"] = self.idx.stage_blocks(blocks)
            b["n_set"] = int(sets.shape[0])
        return b

    def search_requests_per_launch(self, sg):
        return sg.n_get

    def reserve(self, stream):
        self.idx.reserve((1 << self.args.batch_log2) + 8, stream)

    def make_timer(self, slots):
        from megakv_amd.index import SearchTimer
        return SearchTimer(slots)

    def hit_rate(self, b):
        o = b["out"]
        return round(float(((o[:, 0] != 0) | (o[:, 1] != 0)).float().mean().item()), 6)

    def _launch(self, b, stream, pair):
        if not isinstance(pair, tuple) or pair is None:
            # one library call per cycle: memset(out) -> search -> insert; `pair` is
            # (timer handle, slot) packed as a list when the search launch is timed
            timer, slot = (pair[0], pair[1]) if pair else (None, 0)
            self.idx.cycle(b["q"], b["out"], b.get("ptrs"), b.get("cnts"), len(b["keep"]) if "keep" in b else 0,
                           memset=self.args.memset, overlap=self.args.overlap, stream=stream,
                           timer=timer, slot=slot, insert_hint=b.get("n_set", 0))
            return

    def step(self, b, stream, pair):
        if not self.args.graph:
            self._launch(b, stream, pair)
            return
        key = id(b)
        if key not in self.graphs:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=stream):
                self._launch(b, torch.cuda.current_stream(), None)
            self.graphs[key] = gr
        self.graphs[key].replay()

    def table_host(self):
        return self.idx.table_numpy()

    def table_image(self):
        torch.cuda.synchronize()
        return self.idx.table.clone()

    def restore_table(self, image):
        torch.cuda.synchronize()
        self.idx.table.copy_(image)
        torch.cuda.synchronize()

    def step_three_call(self, b, stream):
        """the reference's sequence for one worker and cycle: memset(out), search, insert -- three calls"""
        self.idx.cycle(b["q"], b["out"], b.get("ptrs"), b.get("cnts"), len(b["keep"]) if "keep" in b else 0,
                       memset=True, overlap="none", stream=stream, insert_hint=b.get("n_set", 0))

    def step_abi_literal(self, b, stream):
        """the same through the drop-in symbols an unchanged host links (geometry baked in: MEM_P 30, cuckoo)"""
        import ctypes
        from megakv_amd.index import _dev_ptr, _stream_ptr
        L = self.idx.lib
        n = b["q"].shape[0]
        L.cudaMemsetAsync(_dev_ptr(b["out"]), 0, ctypes.c_size_t(n * 8), _stream_ptr(stream))
        L.gpu_hash_search(_dev_ptr(b["q"]), _dev_ptr(b["out"]), _dev_ptr(self.idx.table), n, 24576, 256, _stream_ptr(stream))
        if "keep" in b:
            L.gpu_hash_insert(_dev_ptr(self.idx.table), _dev_ptr(b["ptrs"]), _dev_ptr(b["cnts"]), len(b["keep"]), _stream_ptr(stream))

    def step_abi_deferred(self, b, stream):
        self.step_abi_literal(b, stream)
        assert self.idx.lib.mkv_flush() == 0

    def prewarm(self, batches, ms, stream):
        """untimed read-only searches of the staged batches into a scratch buffer"""
        scratch = torch.empty_like(batches[0]["out"])
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for b in batches[:16]:
                if b["q"].shape[0] == scratch.shape[0]:
                    self.idx.search(b["q"], out=scratch, stream=stream)
            stream.synchronize()
        return time.perf_counter() - t0

    def verify_sets(self, sets_list, stream):
        """fraction of the given SET requests (host arrays [n,3]) now found under their own
        location -- minimum over the batches"""
        worst = 1.0
        for sets in sets_list:
            if not sets.shape[0]:
                continue
            o = self.idx.search(dev(np.ascontiguousarray(sets[:, :2]), self.device), stream=stream)
            stream.synchronize()
            res = torch.where(o[:, 0] != 0, o[:, 0], o[:, 1]).cpu().numpy().view(np.uint32)
            worst = min(worst, float((res == sets[:, 2]).mean()))
        return round(worst, 6)

    def count_entries(self, dist=None):
        """occupied slots (signature != 0) of the table"""
        return int((self.idx.table.view(-1, 16)[:, :8] != 0).sum().item())


def scheduler_regime(engine, stream, workers=16, ns=7000, ni=6000, nd=1000, cycles=48):
    """Side measurement, never `value`: the regime the reference's caller runs in -- per cycle and worker a few
    thousand searches, inserts (8 blocks) and deletes, 16 workers (src/mega.c:107,136,143-144,163-165;
    src/mega_scheduler.c:393-504) -- as ONE mkv_cycle_workers call per cycle on a table of its own (a quarter of the
    benchmark's: MEM_P - 2), requests resident in HBM.  `enqueued`: the cycles back to back on one stream, one wait at
    the end -- the device's time per cycle; `synced`: a host wait after every cycle, as the scheduler has it
    (cudaDeviceSynchronize, :504) -- launch and wake-up latency included.  Results of the last cycle's searches
    are checked (every key inserted a cycle earlier comes back with its location)."""
    import ctypes
    from megakv_amd.index import _dev_ptr, _stream_ptr
    from megakv_amd.lib import MkvWorker
    g = M.Geometry(mem_p=max(engine.idx.geom.mem_p - 2, 20), scheme=engine.idx.geom.scheme)
    idx = M.HashIndex(g, lib=engine.idx.lib)
    L = idx.lib
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
    key = 1
    work, prev = [], None
    for c in range(cycles):
        row, cur = [], []
        for w in range(workers):
            k = np.arange(key, key + ni, dtype=np.uint64)
            key += ni
            z = (k * np.uint64(0x9E3779B97F4A7C15)) ^ (k >> np.uint64(29))
            z = (z ^ (z >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93)
            sig = (z & np.uint64(0xffffffff)).astype(np.uint32)
            sig[sig == 0] = 1
            hsh = (z >> np.uint64(32)).astype(np.uint32)
            e = np.stack([sig, hsh, k.astype(np.uint32)], axis=1)
            region = (hsh >> np.uint32(29)).astype(np.int64)          # the receiver's binning, src/mega_recv.c:476-477
            blocks = [dev(e[region == b].reshape(-1, 3)) for b in range(8)]
            ptrs, cnts, keep = idx.stage_blocks(blocks)
            src = prev[w] if prev is not None else e                 # search what the worker inserted a cycle ago ...
            pick = np.arange(ns) % src.shape[0]
            q = dev(src[pick, :2])
            out = torch.empty((ns, 2), dtype=torch.int32, device="cuda")
            d = dev(src[-nd:]) if (nd and prev is not None) else None  # ... and delete the last of it (behind the search)
            row.append((q, out, d, ptrs, cnts, keep, src[pick, 2].copy()))
            cur.append(e)
        work.append(row)
        prev = cur
    torch.cuda.synchronize()

    # the argument arrays of every cycle, built ahead (a C host has them in its worker structures)
    args_of = []
    for c in range(cycles):
        ws = (MkvWorker * workers)()
        for w in range(workers):
            q, out, d, ptrs, cnts, keep, _ = work[c][w]
            ws[w].search_in, ws[w].search_out, ws[w].num_search = _dev_ptr(q).value, _dev_ptr(out).value, ns
            ws[w].delete_in, ws[w].num_delete = (_dev_ptr(d).value if d is not None else None), (nd if d is not None else 0)
            ws[w].blk_input, ws[w].blk_elem_num, ws[w].num_blks = _dev_ptr(ptrs).value, _dev_ptr(cnts).value, 8
        args_of.append(ws)
    gref, tptr, sptr = ctypes.byref(idx._g), _dev_ptr(idx.table), _stream_ptr(stream)

    def one(c):
        rc = L.mkv_cycle_workers(gref, tptr, args_of[c], workers, sptr)
        if rc != 0:
            raise RuntimeError(f"mkv_cycle_workers: {rc}")

    out = {}
    for mode in ("enqueued", "synced"):
        idx.table.zero_()
        torch.cuda.synchronize()
        for c in range(4):
            one(c)
        stream.synchronize()
        t0 = time.perf_counter()
        for c in range(4, cycles):
            one(c)
            if mode == "synced":
                stream.synchronize()
        stream.synchronize()
        out[mode] = (time.perf_counter() - t0) / (cycles - 4)
    # the last cycle's searches: keys inserted a cycle earlier, none of them deleted before that search
    wrong = 0
    for w in range(workers):
        q, o, d, ptrs, cnts, keep, want = work[cycles - 1][w]
        res = o.cpu().numpy().view(np.uint32)
        wrong += int(np.count_nonzero(np.where(res[:, 0] != 0, res[:, 0], res[:, 1]) != want))
    ok = wrong == 0
    ops = workers * (ns + ni + nd)
    alg = workers * (ns * 144 + ni * 84 + nd * 80)                  # bytes per cycle: 144 per GET, ~84 per SET, ~80 per delete
    del idx
    return {"what": f"{workers} workers x ({ns} searches, {ni} inserts in 8 blocks, {nd} deletes) per cycle, one mkv_cycle_workers call a cycle, "
                    f"MEM_P={g.mem_p}, requests resident in HBM",
            "us_per_cycle_enqueued": round(out["enqueued"] * 1e6, 1), "mops_enqueued": round(ops / out["enqueued"] / 1e6, 1),
            "us_per_cycle_synced": round(out["synced"] * 1e6, 1), "mops_synced": round(ops / out["synced"] / 1e6, 1),
            "algorithmic_bytes_per_cycle": alg,
            "frac_of_hbm_peak": round(alg / out["enqueued"] / 8e12, 4),
            "bound": "latency: seven dependent launches of a few microseconds of memory traffic each",
            "searches_correct": ok, "searches_wrong": wrong}


def pcie_inclusive(engine, batches, stream, args):
    """Side measurement, never `value`: the same cycles when the boundary hands over HOST
    buffers the way the reference's scheduler does (pinned staging, H2D of the requests, D2H of
    the results, src/mega_scheduler.c:404-419,484-494).
      value        the library's three-stage pipeline (mkv_pipe_*, megakv_amd/csrc/mkv_pipe.hip):
                   copy-in of batch i+1, kernels of batch i and copy-out of batch i-1 on three
                   streams, three sets of device buffers -- the reference's triple buffering
                   (src/include/mega_batch.h:74-97) carried down to the device
      resolved     the same with the results resolved on the device (4 bytes per GET leave it)
      one_stream   copy, kernels, copy on ONE stream as the reference's worker issues them
    Every host result is compared with the device-resident run's (bit for bit).  The batches are
    re-run (their SET keys are then updates), which does not change the bytes moved."""
    import ctypes
    if len(batches) < 8:
        return None
    idx = engine.idx
    L = idx.lib
    n = len(batches)
    keep = []
    pipe = None

    def pinned(nbytes, dtype, shape):
        ptr = L.mkv_host_alloc(max(int(nbytes), 64))
        if not ptr:
            raise MemoryError("mkv_host_alloc")
        keep.append(ptr)
        buf = (ctypes.c_char * max(int(nbytes), 64)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape), ptr

    try:
        # the device-resident results of these batches against the table as it is now
        with torch.cuda.stream(stream):
            for b in batches:
                engine._launch(b, stream, None)
            stream.synchronize()
        host = []
        for b in batches:
            q = b["q"].cpu().numpy()
            hq, pq = pinned(q.nbytes, np.int32, q.shape)
            hq[...] = q
            ho, po = pinned(q.shape[0] * 8, np.uint32, (q.shape[0], 2))
            hl, pl = pinned(q.shape[0] * 4, np.uint32, (q.shape[0],))
            h = {"q": hq, "pq": pq, "out": ho, "po": po, "loc": hl, "pl": pl, "n_get": q.shape[0],
                 "cnts": np.zeros(8, np.int32), "ps": None, "n_set": 0}
            if "sets_all" in b:
                sa = b["sets_all"].cpu().numpy()
                hs, ps = pinned(sa.nbytes, np.int32, sa.shape)
                hs[...] = sa
                h["ps"], h["n_set"] = ps, sa.shape[0]
                h["cnts"] = np.array([k.shape[0] for k in b["keep"]], np.int32)
            host.append(h)
        n_get_max = max(h["n_get"] for h in host)
        n_set_max = max(h["n_set"] for h in host)
        nblk = len(batches[0]["keep"]) if "keep" in batches[0] else 8
        pipe = ctypes.c_void_p(0)
        rc = L.mkv_pipe_create(ctypes.byref(pipe), ctypes.byref(idx._g), ctypes.c_void_p(idx.table.data_ptr()),
                               n_get_max, n_set_max, nblk, 3)
        if rc != 0:
            raise RuntimeError(f"mkv_pipe_create failed ({rc}): {L.mkv_pipe_error(pipe)}")
        P, I64 = ctypes.c_void_p, ctypes.c_int64
        warm = 4
        laps = max(1, 64 // (n - warm))

        def run(sel, resolve):
            m = len(sel)
            gets = (P * m)(*[h["pq"] for h in sel])
            ng = (I64 * m)(*[h["n_get"] for h in sel])
            sets = (P * m)(*[h["ps"] for h in sel])
            cnts = (P * m)(*[h["cnts"].ctypes.data for h in sel])
            outs = (P * m)(*[(h["pl"] if resolve else h["po"]) for h in sel])
            rc = L.mkv_pipe_run(pipe, m, gets, ng, sets, cnts, outs, 1 if resolve else 0,
                                ctypes.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(f"mkv_pipe_run failed ({rc}): {L.mkv_pipe_error(pipe)}")

        res = {}
        for resolve in (0, 1):
            run(host[:warm], resolve)
            stream.synchronize()
            best = None
            for _ in range(3):
                for h in host:
                    (h["loc"] if resolve else h["out"]).fill(0xEE)
                t0 = time.perf_counter()
                run(host[warm:] * laps, resolve)        # (several laps: the pipeline's fill and drain thin out)
                stream.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            # host-side parity: what came back over PCIe equals the device-resident run's results
            ok = True
            for b, h in list(zip(batches, host))[warm:]:
                dev_out = b["out"].cpu().numpy().view(np.uint32)
                if resolve:
                    ok = ok and bool(np.array_equal(h["loc"], np.where(dev_out[:, 0] != 0, dev_out[:, 0], dev_out[:, 1])))
                else:
                    ok = ok and bool(np.array_equal(h["out"], dev_out))
            steps = (n - warm) * laps
            res[resolve] = {"value": round(steps * (1 << args.batch_log2) / best / 1e6, 1), "unit": "Mops",
                            "us_per_step": round(best / steps * 1e6, 1), "host_results_identical": ok}
        L.mkv_pipe_destroy(pipe)
        pipe = None
        # the reference's own issue order: one stream, no overlap
        hq = [torch.from_numpy(h["q"]) for h in host]
        t_host = [{"q": b["q"].cpu().pin_memory(), "out": torch.empty_like(b["out"], device="cpu").pin_memory(),
                   "sets": b["sets_all"].cpu().pin_memory() if "sets_all" in b else None} for b in batches[:10]]
        del hq

        def serial(pairs):
            for b, h in pairs:
                b["q"].copy_(h["q"], non_blocking=True)
                if h["sets"] is not None:
                    b["sets_all"].copy_(h["sets"], non_blocking=True)
                engine._launch(b, stream, None)
                h["out"].copy_(b["out"], non_blocking=True)
            stream.synchronize()

        with torch.cuda.stream(stream):
            serial(list(zip(batches, t_host))[:4])
            ds = None
            for _ in range(3):            # (now and then a whole pass of pinned copies takes ten times as long here)
                t0 = time.perf_counter()
                serial(list(zip(batches, t_host))[4:10])
                d1 = (time.perf_counter() - t0) / 6
                ds = d1 if ds is None else min(ds, d1)
        out = dict(res[0])
        out.update({
            "steps": (n - warm) * laps, "stages": "copy-in | mkv_cycle | copy-out on three streams, three sets of device buffers",
            "resolved": res[1],
            "one_stream": {"value": round((1 << args.batch_log2) / ds / 1e6, 1), "unit": "Mops",
                           "us_per_step": round(ds * 1e6, 1)},
            "bytes_per_step_h2d": int(host[0]["n_get"] * 8 + host[0]["n_set"] * 12),
            "bytes_per_step_d2h": int(host[0]["n_get"] * 8),
            "note": "pinned host buffers; H2D requests + D2H results per step; best of three passes"})
        return out
    finally:
        # the pipeline first (it drains the device): copies on the pinned buffers may be in flight
        if pipe is not None and pipe.value:
            L.mkv_pipe_destroy(pipe)
        torch.cuda.synchronize()
        for ptr in keep:
            L.mkv_host_free(ptr)


def library_id(engine):
    try:
        return engine.idx.lib.mkv_version().decode().split()[-1]
    except Exception:
        return None


def profiler_present():
    """is this process itself running under a profiler (rocprofv3's tool library in the environment)?"""
    keys = ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH")
    return any(os.environ.get(k) for k in keys) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def counter_per_launch(path, kern, counter, warm, steps):
    """rocprofv3's counter_collection.csv -> (mean value of `counter` per dispatch of the kernel whose name contains
    `kern`, dispatches averaged, dispatches seen): a dispatch's rows (one per instance the counter is kept in) are
    summed; the dispatches are taken in order, the first `warm` (the warm-up steps) dropped and the next `steps` (the
    timed steps; behind them the run checks its results with further searches) averaged.  (None, 0, seen) when fewer
    than half of `steps` are there."""
    import csv
    per = {}
    for row in csv.DictReader(open(path)):
        if kern in row["Kernel_Name"] and row["Counter_Name"] == counter:
            d = int(row["Dispatch_Id"])
            per[d] = per.get(d, 0.0) + float(row["Counter_Value"])
    ids = sorted(per)[warm:warm + steps]
    if len(ids) < max(1, steps // 2):
        return None, 0, len(per)
    return sum(per[d] for d in ids) / len(ids), len(ids), len(per)


def live_traffic(args, fused):
    """HBM bytes per launch of the dominant kernel, measured by THIS run: two child processes of this script (the same
    workload flags, 5 warm-up + 20 steps once, no side legs) under `rocprofv3 --kernel-trace --pmc <counter>`, FETCH_SIZE
    and WRITE_SIZE each in a pass of its own and with no other trace domain, as MI355X_MICROARCH.md's HBM / rocprofv3
    section prescribes; the counters are in KB, averaged over the kernel's dispatches behind the warm-up.  Returns
    (bytes per launch, note) or (None, why not).  The children are started with subprocess -- nothing is exec'ed from
    this process, which holds the GPU -- and each has 150 s."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "no rocprofv3 on this machine"
    if profiler_present():
        return None, "this run is itself under a profiler"
    kern = "k_search_mark" if fused else "k_search<"
    warm, steps = 5, 20
    # (the interpreter this run runs under, by its real path: a "python3" looked up on PATH may be another one without
    # torch, or a wrapper script that execs the real one behind the profiler's preloaded library -- which this pool forbids)
    child = [os.path.realpath(sys.executable), os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warm), "--repeats", "1",
             "--workload", args.workload, "--keying", args.keying, "--sets", args.sets, "--scheme", args.scheme,
             "--lf", str(args.lf), "--batch-log2", str(args.batch_log2), "--mem-p", str(args.mem_p), "--overlap", args.overlap,
             "--no-cpu", "--no-pcie", "--no-abi", "--no-sched", "--no-events", "--no-live-traffic", "--prewarm-ms", "0"]
    if args.memset:
        child.append("--memset")
    if args.sorted_gets:
        child.append("--sorted-gets")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got, launches = {}, 0
    top = tempfile.mkdtemp(prefix="mkv_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(top, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + child
            try:
                r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=150)
            except subprocess.TimeoutExpired:
                return None, f"the {counter} pass did not finish in 150 s"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                tail = (r.stderr or b"").decode(errors="replace").strip().splitlines()[-1:]
                return None, f"the {counter} pass failed (exit {r.returncode}{': ' + tail[0][:120] if tail else ''})"
            mean, launches, seen = counter_per_launch(max(files, key=os.path.getmtime), kern, counter, warm, steps)
            if mean is None:
                return None, f"the {counter} pass shows {seen} dispatches of {kern}"
            got[counter] = mean
    finally:
        shutil.rmtree(top, ignore_errors=True)
    total = int((got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)
    return total, (f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, a child pass each of this run's workload "
                   f"({launches} launches of {kern.rstrip('<')} behind {warm} warm-up steps averaged; "
                   f"FETCH {got['FETCH_SIZE']:.0f} KB + WRITE {got['WRITE_SIZE']:.0f} KB)")


def load_traffic(args, fused, sharded):
    """HBM bytes per launch of the dominant kernel: NOT measured by this run -- read from the
    committed rocprofv3 PMC passes of this command (tools/make_profiles.py writes the file,
    separate --pmc passes, FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes) and
    labelled with where it came from; null when there is no entry for this kernel/workload."""
    p = os.path.join(ROOT, "profiles", "search_traffic.json")
    try:
        d = json.load(open(p))
        kern = "k_search_slots_mark" if sharded else ("k_search_mark" if fused else "k_search")
        e = d.get(f"{args.workload}-{args.keying}-{kern}")
        if not e:
            return None, None, None
        return e.get("hbm_bytes_per_launch"), f"profiles/{e.get('source', 'search_traffic.json')}@{e.get('head', '?')}", e.get("library")
    except Exception:
        return None, None, None


def step_algorithmic(sg, cpu, s_per_step):
    """Whole-step figure: (GETs x 144 B + SETs x oracle-counted bytes) / step time.  SET bytes =
    12 + 64 V + 4 W, V buckets visited and W dword stores counted by the CPU
    oracle on the cpu_baseline sample of the same workload; 84 B (V=1, W=2) without it."""
    set_b, src = 84.0, "assumed 84 B per SET (no oracle sample in this run)"
    if isinstance(cpu, dict) and cpu.get("set_bytes_per_request"):
        set_b, src = float(cpu["set_bytes_per_request"]), "oracle-counted on the cpu_baseline sample"
    total = GET_BYTES * sg.n_get + set_b * sg.n_set
    gbps = total / s_per_step / 1e9
    return {"bytes_per_step": int(total), "GBps": round(gbps, 1), "frac_of_peak": round(gbps / HBM_PEAK_GBPS, 4),
            "set_bytes_per_request": round(set_b, 2), "set_bytes_source": src, "per_rank": True}


def cpu_baseline(engine, sg, args):
    """The CPU oracle (a port: the reference has no CPU index path) on the
    host's cores, same table image, a bounded sample of the same workload."""
    from oracle import oracle as O
    native = O.use_native()           # -O3 -march=native, built on this box
    scheme = O.SCHEME_CUCKOO if args.scheme == "cuckoo" else O.SCHEME_2CHOICE
    og = O.geom(mem_p=args.mem_p, scheme=scheme)
    # the table image in transparent huge pages where the host allows it (madvise): with 4 KiB
    # pages every probe of a 1 GiB table is a TLB miss as well -- a careful CPU implementation
    # would not leave that on the table (the oracle's loops also prefetch a dozen requests ahead)
    src = engine.table_host()
    table, keep, huge = None, None, False
    try:
        import mmap
        keep = mmap.mmap(-1, src.nbytes)
        if hasattr(mmap, "MADV_HUGEPAGE"):
            keep.madvise(mmap.MADV_HUGEPAGE)
            huge = True
        table = np.frombuffer(keep, dtype=np.uint32).reshape(src.shape)
        np.copyto(table, src)
    except Exception:                                             # noqa: BLE001
        table, huge = np.ascontiguousarray(src), False
    del src
    nb = max(1, args.cpu_batches)
    work = [sg.batch() for _ in range(nb)]
    g = engine.idx.geom
    t0 = time.perf_counter()
    ops = 0
    nset = vis = stores = 0
    for gets, sets in work:
        out = np.zeros((gets.shape[0], 2), np.uint32)           # the memset
        O.search(og, table, gets, out=out)
        if sets.shape[0]:
            st = O.insert(og, table, W.bin_inserts(sets, g.mem_p, g.iblock_p, W.BIN_REGION))
            nset += st["elems"]
            vis += st["buckets_visited"]
            stores += st["dword_stores"]
        ops += gets.shape[0] + sets.shape[0]
    dt = time.perf_counter() - t0
    one = ops / dt / 1e6
    # all host cores: searches split over the threads (read-only), inserts one thread per closed
    # region (regions never share a bucket) -- the same batches once more
    ncpu = int(os.environ.get("MKV_CPU_THREADS", "0")) or min(os.cpu_count() or 1, 64)
    t0 = time.perf_counter()
    t_search = 0.0
    for gets, sets in work:
        ts = time.perf_counter()
        O.search(og, table, gets, threads=ncpu)
        t_search += time.perf_counter() - ts
        if sets.shape[0]:
            O.insert_mt(og, table, W.bin_inserts(sets, g.mem_p, g.iblock_p, W.BIN_REGION))
    dt_mt = time.perf_counter() - t0
    mt_search = sum(w[0].shape[0] for w in work) / max(t_search, 1e-9) / 1e6
    return {"value": round(one, 2), "unit": "Mops", "cores": 1, "kind": "port",
            "set_bytes_per_request": round(12 + (64 * vis + 4 * stores) / nset, 2) if nset else None,
            "sample": f"{nb} batches of 2^{args.batch_log2} requests of the same workload, same table image, "
                      f"single thread in canonical order ({dt:.1f} s); software prefetch 12 requests ahead, "
                      f"table in {'transparent huge pages (madvise)' if huge else '4 KiB pages'}; "
                      f"built {'-O3 -march=native on this box' if native else '-O3 (portable)'}",
            "all_threads": {"value": round(ops / dt_mt / 1e6, 1), "threads": ncpu,
                            "insert_threads": 1 << g.iblock_p,
                            "note": "searches over a persistent pool of host threads, inserts one thread per closed "
                                    "region (64 threads: 128 or 256 of this box's 256 hardware threads run slower)"},
            "search_only_all_threads": {"value": round(mt_search, 1), "threads": ncpu}}


if __name__ == "__main__":
    main()
