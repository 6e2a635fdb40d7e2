"""Sharded (multi-GPU) hash index: one process per GPU, RCCL all-to-all over xGMI.

New design -- the reference is single-GPU.  The table shards
naturally: an entry's alternate bucket keeps the top bits of its primary bucket
index (reference libgpuhash/gpu_hash.cu:66-67), so with G = 2^k ranks and an
aggregate table of MEM_P_agg = MEM_P_local + k, rank g owns the buckets whose
top k index bits are g, as a local MEM_P_local table whose region split is
IBLOCK_P - k (the alternate rule restricted to the shard).  A request is served
entirely by shard(hash) = (hash & HASH_MASK_agg) >> (MEM_P_local - 6).

One step on every rank (all stream-ordered, no host synchronisation):
  1. pack     local batch -> G equal slots [n_get, n_set | gets | sets], stable
              (mkv_pack_requests, three short kernels)
  2. forward  all_to_all_single of the slots (equal splits: no size exchange)
  3. local    search of every received slot (counts read from the headers on the
              device), then insert with one block per source rank -- canonical
              order inside a shard = (source rank, index in its batch)
  4. back     all_to_all_single of the result slots
  5. unpack   results to the original request order

The collective is injected (`comm`): RCCL through torch.distributed for real
runs, a CPU-staged gloo exchange so two processes can share one GPU in tests.
The device operations are injected too (`ops`), so the orchestration can run on
CPU tensors in the gloo tests.
"""
import ctypes
import math

import numpy as np
import torch

from .index import Geometry, HashIndex, _dev_ptr, _stream_ptr, _check


def shard_geometry(mem_p_local, world, scheme, iblock_p=3, max_cuckoo=5):
    k = world.bit_length() - 1
    assert 1 << k == world, "world size must be a power of two"
    assert k <= iblock_p, "at most 2^IBLOCK_P shards: regions are the closed units"
    return Geometry(mem_p=mem_p_local, scheme=scheme, iblock_p=iblock_p - k, max_cuckoo=max_cuckoo)


class NcclComm:
    """RCCL all-to-all on device tensors (stream-ordered by torch.distributed)."""

    def __init__(self, dist):
        self.dist = dist

    def all_to_all(self, out, inp):
        self.dist.all_to_all_single(out, inp)


class StagedComm:
    """all-to-all through host memory with any backend (gloo): lets several
    ranks share one GPU in tests.  Synchronises the stream; not for benchmarks."""

    def __init__(self, dist):
        self.dist = dist

    def all_to_all(self, out, inp):
        torch.cuda.current_stream().synchronize() if inp.is_cuda else None
        h_in = inp.cpu()
        h_out = torch.empty_like(h_in)
        self.dist.all_to_all_single(h_out, h_in)
        out.copy_(h_out)


class HipOps:
    """The device side of a step, through the C ABI of the HIP library."""

    def __init__(self, index: HashIndex):
        self.idx = index
        self.lib = index.lib

    def slot_words(self, cap_get, cap_set):
        return int(self.lib.mkv_pack_slot_words(cap_get, cap_set))

    def scratch_bytes(self, n_get, n_set, g):
        return int(self.lib.mkv_pack_scratch_bytes(n_get, n_set, g))

    @staticmethod
    def _sp(stream):
        return _stream_ptr(None) if stream is None else ctypes.c_void_p(stream)

    def pack(self, gets, sets, agg_mask, shift, g, sendbuf, cap_get, cap_set, perm, scratch, overflow, stream=None):
        null = ctypes.c_void_p(0)
        _check(self.lib.mkv_pack_requests(
            _dev_ptr(gets) if gets.shape[0] else null, gets.shape[0],
            _dev_ptr(sets) if sets.shape[0] else null, sets.shape[0],
            agg_mask, shift, g, _dev_ptr(sendbuf), cap_get, cap_set, _dev_ptr(perm), _dev_ptr(scratch),
            scratch.numel() * scratch.element_size(), _dev_ptr(overflow), self._sp(stream)), "mkv_pack_requests")

    def search_slots(self, recvbuf, g, cap_get, cap_set, out_slots, set_counts, stream=None):
        _check(self.lib.mkv_search_slots(ctypes.byref(self.idx._g), _dev_ptr(recvbuf), g, cap_get, cap_set,
                                         _dev_ptr(out_slots), _dev_ptr(self.idx.table), _dev_ptr(set_counts),
                                         self._sp(stream)), "mkv_search_slots")

    def insert_slots(self, ptrs, set_counts, g, hint=0, stream=None):
        self.idx.insert_staged(ptrs, set_counts, g, stream, hint)

    def slots_cycle(self, recvbuf, g, cap_get, cap_set, out_slots, set_ptrs, hint, stream=None, timer=None, slot=0):
        """search of the received slots + the inserts they carry, one library call"""
        null = ctypes.c_void_p(0)
        _check(self.lib.mkv_slots_cycle(ctypes.byref(self.idx._g), _dev_ptr(recvbuf), g, cap_get, cap_set,
                                        _dev_ptr(out_slots), _dev_ptr(self.idx.table),
                                        _dev_ptr(set_ptrs) if cap_set else null, int(hint), self._sp(stream),
                                        ctypes.c_void_p(timer) if timer else null, slot), "mkv_slots_cycle")

    def unpack(self, back, perm, sendbuf, g, cap_get, cap_set, out, stream=None):
        _check(self.lib.mkv_unpack_results(_dev_ptr(back), _dev_ptr(perm), out.shape[0], _dev_ptr(out),
                                           self._sp(stream)), "mkv_unpack_results")

    def slot_set_pointers(self, recvbuf, g, slot_words, cap_get):
        base = recvbuf.data_ptr()
        return torch.tensor([base + 4 * (s * slot_words + 4 + 2 * cap_get) for s in range(g)],
                            dtype=torch.int64).to(recvbuf.device)


class NativeEngine:
    """One rank's sharded step driven by the C library (mkv_shard_*): pack, RCCL
    all-to-all, local search+insert, result exchange, unpack -- a whole run of batches,
    two in flight, enqueued by ONE library call.  The communicator is the library's own;
    `dist` (torch.distributed, any backend) only carries the 128-byte id to the ranks."""

    def __init__(self, sh, dist, loopback=False):
        """loopback: every peer is this process (the exchanges are device copies) -- the W-rank
        schedule on one GPU, for tests (mkv_shard.hip, loopback_api; the diagnostics build only:
        ShardedIndex(..., lib=load_library(diagnostics=True)))."""
        import os
        self.sh, self.lib = sh, sh.idx.lib
        self.h = ctypes.c_void_p(0)
        # (MKV_RCCL_PATH: another RCCL than the one the host framework loaded -- tests point it nowhere to see
        # the set-up fail loudly)
        rccl = os.environ.get("MKV_RCCL_PATH") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = b"loopback" if loopback else (rccl.encode() if (os.path.exists(rccl) or "MKV_RCCL_PATH" in os.environ) else None)
        rc = self.lib.mkv_shard_create(ctypes.byref(self.h), ctypes.byref(sh.idx._g), sh.world, sh.rank,
                                       _dev_ptr(sh.idx.table), sh.n_get, sh.n_set, sh.cap_get, sh.cap_set,
                                       path)
        self._check(rc, "mkv_shard_create")
        ident = [None]
        if sh.rank == 0:
            buf = ctypes.create_string_buffer(128)
            self._check(self.lib.mkv_shard_unique_id(self.h, buf), "mkv_shard_unique_id")
            ident = [buf.raw]
        if sh.world > 1 and not loopback:
            dist.broadcast_object_list(ident, src=0)
        self._check(self.lib.mkv_shard_connect(self.h, ctypes.create_string_buffer(ident[0], 128)),
                    "mkv_shard_connect")

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.mkv_shard_error(self.h) if self.h else b""
            text = (msg or b"").decode(errors="replace")
            if what in ("mkv_shard_create", "mkv_shard_unique_id", "mkv_shard_connect"):
                self.close()                      # a half-built engine is of no use: free what it holds
            raise RuntimeError(f"{what} failed ({rc}): {text}")

    def prepare(self, batches, timer=None, timer_every=0):
        """The argument arrays of a run (device pointers and counts) built once; returns a callable
        that enqueues the run on the stream current at call time."""
        n = len(batches)
        if n == 0:
            return lambda: None
        P, I64 = ctypes.c_void_p, ctypes.c_int64
        gets = (P * n)(*[b["q"].data_ptr() if b["q"].shape[0] else None for b in batches])
        sets = (P * n)(*[b["s"].data_ptr() if b["s"].shape[0] else None for b in batches])
        outs = (P * n)(*[b["out"].data_ptr() for b in batches])
        ng = (I64 * n)(*[b["q"].shape[0] for b in batches])
        ns = (I64 * n)(*[b["s"].shape[0] for b in batches])
        tm = ctypes.c_void_p(timer) if timer else None

        def go(stream=None):
            self._check(self.lib.mkv_shard_run(self.h, n, gets, ng, sets, ns, outs, _stream_ptr(stream), tm,
                                               int(timer_every)), "mkv_shard_run")
        return go

    def run(self, batches, stream=None, timer=None, timer_every=0):
        self.prepare(batches, timer, timer_every)(stream)

    def overflowed(self):
        flag = ctypes.c_int(0)
        self._check(self.lib.mkv_shard_overflowed(self.h, ctypes.byref(flag)), "mkv_shard_overflowed")
        return bool(flag.value)

    def phases_us(self, batches, stream=None, traced=8):
        """One more run of `batches` with device timestamps around every operation of `traced` of them: mean
        microseconds of pack / exchange / local / un-pack (mkv_shard_trace; a pass of its own -- the events
        cost time -- and only with more than traced + 16 batches)."""
        self.lib.mkv_shard_trace(int(traced))
        try:
            self.run(batches, stream)
        finally:
            out, cnt = (ctypes.c_float * 4)(), (ctypes.c_int * 4)()
            self.lib.mkv_shard_phases(out, cnt)
        return {k: round(float(out[i]), 2) for i, k in enumerate(("pack", "exchange", "local", "unpack"))} | \
            {"spans": int(cnt[2])}

    def close(self):
        if self.h:
            self.lib.mkv_shard_destroy(self.h)
            self.h = ctypes.c_void_p(0)


class _TorchTimer:
    """event pairs around the local search launch (bench.py roofline at N > 1)"""

    def __init__(self, slots):
        self.h = self
        self.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                   for _ in range(slots)]

    def start(self, slot):
        self.ev[slot][0].record()

    def stop(self, slot):
        self.ev[slot][1].record()

    def elapsed_ms(self):
        return [a.elapsed_time(b) for a, b in self.ev]


class _Lane:
    """One set of exchange buffers + a stream: two lanes let the collectives of one
    batch overlap the local kernels of the neighbouring batch."""

    def __init__(self, sh, device, make_stream):
        w, ops = sh.world, sh.ops
        i32 = dict(dtype=torch.int32, device=device)
        self.sendbuf = torch.zeros((w, sh.slot_words), **i32)
        self.recvbuf = torch.zeros((w, sh.slot_words), **i32)
        self.out_slots = torch.zeros((w, max(sh.cap_get, 1), 2), **i32)
        self.back_slots = torch.zeros((w, max(sh.cap_get, 1), 2), **i32)
        self.perm = torch.zeros((w * max(sh.cap_get, 1),), **i32)
        self.scratch = torch.zeros((ops.scratch_bytes(max(sh.n_get, 1), max(sh.n_set, 1), w) // 4 + 16,), **i32)
        self.set_counts = torch.zeros((w,), **i32)
        self.set_ptrs = ops.slot_set_pointers(self.recvbuf, w, sh.slot_words, sh.cap_get)
        self.stream = make_stream() if make_stream else None
        self.sptr = self.stream.cuda_stream if self.stream is not None else None
        if self.stream is not None:
            self.ev_local = [torch.cuda.Event(), torch.cuda.Event()]   # alternated so a pending wait keeps its event
            self.ev_done = torch.cuda.Event()


class ShardedIndex:
    def __init__(self, mem_p_local, world, rank, scheme, device, n_get, n_set, comm, ops=None,
                 cap_factor=None, index=None, lib=None):
        # (lib: another build of the library than the product's -- the looped-back peers of the tests and
        # probes exist in the diagnostics build only, load_library(diagnostics=True))
        self.world, self.rank = world, rank
        self.native = None
        self.device = torch.device(device)
        self.geom = shard_geometry(mem_p_local, world, scheme)
        self.k = world.bit_length() - 1
        self.mem_p_agg = mem_p_local + self.k
        self.agg_mask = (1 << (self.mem_p_agg - 6)) - 1
        self.shift = mem_p_local - 6
        self.comm = comm
        if ops is None:
            index = index or HashIndex(self.geom, device=device, lib=lib)
            ops = HipOps(index)
        self.idx = index
        self.ops = ops
        # Slot capacity over the average share.  Largest share of one shard under the benchmark's
        # Zipf-0.99 GET streams (K-mix keying, all ranks, 130 batches each, tools/shard_share.py):
        # 1.072x / 1.098x / 1.243x the average at G = 2 / 4 / 8 -- the hottest key alone is
        # ~4.6 % of the GETs and lands on one shard.  Padding is exchanged like payload, so the
        # margins are kept small; an overflow raises a device flag checked at the end of a run.
        if cap_factor is None:
            cap_factor = {1: 1.0, 2: 1.15, 4: 1.2, 8: 1.4}.get(world, 2.0)
        self.cap_factor = cap_factor
        self.n_get, self.n_set = n_get, n_set
        self.cap_get = int(math.ceil(n_get / world * cap_factor / 256.0)) * 256 if n_get else 0
        self.cap_set = int(math.ceil(n_set / world * cap_factor / 16.0)) * 16 + 16 if n_set else 0
        self.slot_words = ops.slot_words(self.cap_get, self.cap_set)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=self.device)
        on_gpu = self.device.type == "cuda"
        mk = (lambda: torch.cuda.Stream(device=self.device)) if on_gpu else None
        self.lanes = [_Lane(self, self.device, mk), _Lane(self, self.device, mk)]
        # attributes of lane 0 under their old names (single-lane step_tensors)
        for name in ("sendbuf", "recvbuf", "out_slots", "back_slots", "perm", "scratch", "set_counts", "set_ptrs"):
            setattr(self, name, getattr(self.lanes[0], name))

    # -- shard of a request ---------------------------------------------------------
    def shard_of(self, hash_np):
        h = np.ascontiguousarray(hash_np, dtype=np.uint32)       # (a strided column is several times slower)
        return (h & np.uint32(self.agg_mask)) >> np.uint32(self.shift)

    # -- one step -----------------------------------------------------------------------
    def _forward(self, ln, gets, sets, sp=None):
        """pack the local batch into slots and start the forward exchange (sp: raw stream
        handle = the CURRENT torch stream, which the collective uses)"""
        self.ops.pack(gets, sets, self.agg_mask, self.shift, self.world, ln.sendbuf, self.cap_get, self.cap_set,
                      ln.perm, ln.scratch, self.overflow, sp)
        self.comm.all_to_all(ln.recvbuf, ln.sendbuf)

    def _local(self, ln, timing=None, sp=None):
        """serve what the peers sent: search every slot, then insert (source-rank order)"""
        o, w = self.ops, self.world
        if hasattr(o, "slots_cycle"):
            o.slots_cycle(ln.recvbuf, w, self.cap_get, self.cap_set, ln.out_slots, ln.set_ptrs, self.n_set, sp,
                          (timing[0] if isinstance(timing[0], int) else timing[0].h) if timing else None,
                          timing[1] if timing else 0)
            return
        o.search_slots(ln.recvbuf, w, self.cap_get, self.cap_set, ln.out_slots, ln.set_counts)
        if self.cap_set:
            o.insert_slots(ln.set_ptrs, ln.set_counts, w, self.n_set)

    def _backward(self, ln, out, sp=None):
        self.comm.all_to_all(ln.back_slots, ln.out_slots)
        self.ops.unpack(ln.back_slots, ln.perm, ln.sendbuf, self.world, self.cap_get, self.cap_set, out, sp)

    def step_tensors(self, gets, sets, out):
        """gets [n,2], sets [m,3] int32 tensors of this rank; out [n,2] receives the
        search results in request order.  Everything is enqueued on the current stream."""
        ln = self.lanes[0]
        self._forward(ln, gets, sets)
        self._local(ln, getattr(self, "_timing", None))
        self._backward(ln, out)

    def prepare_pipelined(self, batches, timing_of=None):
        """run_pipelined with the marshalling done now: returns a callable that enqueues the run."""
        if self.native is None or not batches:
            return lambda: self.run_pipelined(batches, timing_of) if batches else None
        timer, every = None, 0
        if timing_of is not None:
            # (timing_of(i) -> [timer handle, slot] on every `every`-th batch, slot = i // every)
            hits = [i for i in range(len(batches)) if timing_of(i)]
            if hits:
                timer = timing_of(hits[0])[0]
                every = hits[1] - hits[0] if len(hits) > 1 else len(batches)
        return self.native.prepare(batches, timer if isinstance(timer, int) or timer is None else timer.h, every)

    def run_pipelined(self, batches, timing_of=None):
        """All batches, software-pipelined over two lanes: the forward exchange of batch
        i+1 and the result exchange of batch i-1 overlap the local kernels of batch i.
        Local phases stay strictly ordered (event chain), so results are those of the
        batches run one after the other.  Collectives are issued in the same order on
        every rank: fwd(0), fwd(1), back(0), fwd(2), back(1), ...
        batches: dicts with "q", "s", "out".  Enqueues on the lanes' streams; the
        caller's current stream is joined at the end."""
        if not batches:
            return
        if self.native is not None:
            self.prepare_pipelined(batches, timing_of)()
            return
        main = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record(main)
        for ln in self.lanes:
            ln.stream.wait_event(start)
        prev_local = None
        set_stream = torch.cuda.set_stream         # cheaper than the context manager
        l0 = self.lanes[0]
        set_stream(l0.stream)
        self._forward(l0, batches[0]["q"], batches[0]["s"], l0.sptr)
        n = len(batches)
        for i, b in enumerate(batches):
            ln = self.lanes[i & 1]
            if i + 1 < n:
                nxt = self.lanes[(i + 1) & 1]
                set_stream(nxt.stream)
                self._forward(nxt, batches[i + 1]["q"], batches[i + 1]["s"], nxt.sptr)
            set_stream(ln.stream)
            if prev_local is not None:
                ln.stream.wait_event(prev_local)
            self._local(ln, timing_of(i) if timing_of else None, ln.sptr)
            prev_local = ln.ev_local[(i >> 1) & 1]
            prev_local.record(ln.stream)
            self._backward(ln, b["out"], ln.sptr)
        set_stream(main)
        for ln in self.lanes:
            ln.ev_done.record(ln.stream)
            main.wait_event(ln.ev_done)

    def overflowed(self):
        if self.native is not None and self.native.overflowed():
            return True
        return bool(self.overflow.item())

    def enable_native(self, dist, loopback=False):
        """Switch run_pipelined to the C engine (its own RCCL communicator).  Collective:
        call on every rank.  Raises if the library cannot open RCCL or connect."""
        self.native = NativeEngine(self, dist, loopback)
        return self.native
