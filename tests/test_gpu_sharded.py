"""The sharded path with the real HIP kernels: 2 and 4 ranks share the one GPU of
the box (collective = gloo staged through host memory; RCCL needs one GPU per
rank), checked against one aggregate oracle table."""
import pytest

from tests.sharded_check import run_and_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,scheme", [(2, 0), (2, 1), (4, 0)])
def test_sharded_hip_ranks_on_one_gpu(world, scheme):
    run_and_check(world, "gpu", mem_p_local=16, scheme=scheme, seed=20 + world, timeout=900)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_pipelined_two_lanes(world):
    """run_pipelined (two batches in flight, exchanges overlapping local kernels) must
    give exactly the results of the batches run one after the other."""
    run_and_check(world, "gpu-pipe", mem_p_local=16, scheme=0, seed=30 + world, timeout=900)


PACK_THREADS_DEFAULT = 1024


@pytest.mark.parametrize("threads", [1024, 512, 256])
@pytest.mark.parametrize("g,n_get,n_set,three_pass", [(1, 5000, 300, 0), (2, 70001, 4099, 0), (8, 200000, 30000, 0),
                                                        (4, 0, 2500, 0), (4, 3000, 0, 0), (8, 50000, 5000, 1),
                                                        (1, 1 << 20, 52428, 0), (8, 996149, 52427, 0), (16, 600001, 9, 0),
                                                        (2, 1, 1, 0), (8, 4097, 4097, 0), (8, 524288, 524288, 0)])
def test_pack_layout(g, n_get, n_set, three_pass, threads):
    """mkv_pack_requests: every request lands in the slot of its shard; SET slots keep
    batch order (the canonical insert order of a shard is source rank, then index); GET
    slots hold the right requests in any order, `perm[e]` telling where request e went;
    headers count.  Then the way back: results laid out like the slots return to request order."""
    import ctypes
    import numpy as np
    import torch
    from megakv_amd.lib import load_library
    from megakv_amd.index import _dev_ptr, _stream_ptr
    if three_pass and threads != 1024:
        pytest.skip("the three-pass pack has one form")
    L = load_library(diagnostics=bool(three_pass))      # (forcing the fall-back: the diagnostics build's tuning)
    if three_pass:
        assert L.mkv_set_tuning(b"pack_three_pass", 1) == 0
    assert L.mkv_set_tuning(b"pack_threads", threads) == 0
    try:
        rng = np.random.default_rng(g * 1000 + n_get % 97)
        mem_p_agg = 20
        mask, shift = (1 << (mem_p_agg - 6)) - 1, mem_p_agg - 6 - (g.bit_length() - 1)
        gets = rng.integers(0, 2**32, size=(n_get, 2), dtype=np.uint64).astype(np.uint32)
        sets = rng.integers(0, 2**32, size=(n_set, 3), dtype=np.uint64).astype(np.uint32)
        cap_get = max(256, (int(n_get / g * 1.5) + 255) // 256 * 256) if n_get else 0
        cap_set = (int(n_set / g * 1.5) + 15) // 16 * 16 + 16 if n_set else 0
        sw = int(L.mkv_pack_slot_words(cap_get, cap_set))
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
        dg, ds = dev(gets), dev(sets)
        sendbuf = torch.zeros((g, sw), dtype=torch.int32, device="cuda")
        perm = torch.full((max(n_get, 1) + 3,), -1, dtype=torch.int32, device="cuda")
        nbytes = int(L.mkv_pack_scratch_bytes(max(n_get, 1), max(n_set, 1), g))
        scratch = torch.zeros((nbytes // 4 + 16,), dtype=torch.int32, device="cuda")
        overflow = torch.zeros((1,), dtype=torch.int32, device="cuda")
        null = ctypes.c_void_p(0)
        for rep in range(2):          # twice: the kernel leaves its control words clean
            sendbuf.zero_()
            rc = L.mkv_pack_requests(_dev_ptr(dg) if n_get else null, n_get, _dev_ptr(ds) if n_set else null, n_set,
                                     mask, shift, g, _dev_ptr(sendbuf), cap_get, cap_set, _dev_ptr(perm),
                                     _dev_ptr(scratch), scratch.numel() * 4, _dev_ptr(overflow), _stream_ptr(None))
            assert rc == 0
            torch.cuda.synchronize()
            sb = sendbuf.cpu().numpy().view(np.uint32)
            pm = perm.cpu().numpy()
            assert int(overflow.item()) == 0
            gsh = (gets[:, 1] & mask) >> shift
            ssh = (sets[:, 1] & mask) >> shift
            where = pm[:n_get]
            if n_get:
                assert (where >= 0).all() and len(np.unique(where)) == n_get           # every GET has its own place
                assert np.array_equal(where // cap_get, gsh)                            # ... in the slot of its shard
            for s in range(g):
                want_g = np.flatnonzero(gsh == s)
                want_s = np.flatnonzero(ssh == s)
                assert sb[s, 0] == len(want_g) and sb[s, 1] == len(want_s) and sb[s, 2] == 0 and sb[s, 3] == 0
                got_g = sb[s, 4:4 + 2 * cap_get].reshape(-1, 2)
                pos = where[want_g] - s * cap_get
                assert (pos < len(want_g)).all()                                        # packed densely
                assert np.array_equal(got_g[pos], gets[want_g])
                got_s = sb[s, 4 + 2 * cap_get:4 + 2 * cap_get + 3 * len(want_s)].reshape(-1, 3)
                assert np.array_equal(got_s, sets[want_s])
            if n_get:
                # results as the peers would send them back: position p of slot s holds {s * cap_get + p, ~}
                back = np.arange(g * cap_get, dtype=np.uint32)
                back = np.stack([back, ~back], axis=1)
                dback = dev(back)
                for shift_words in (0, 2):                                               # (16-byte aligned results or not)
                    outbuf = torch.full((2 * (n_get + 1) + 2,), -1, dtype=torch.int32, device="cuda")
                    out = outbuf[shift_words:shift_words + 2 * (n_get + 1)].view(n_get + 1, 2)
                    rc = L.mkv_unpack_results(_dev_ptr(dback), _dev_ptr(perm), n_get, _dev_ptr(out), _stream_ptr(None))
                    assert rc == 0
                    torch.cuda.synchronize()
                    o = out.cpu().numpy().view(np.uint32)
                    assert np.array_equal(o[:n_get], back[where]), "un-pack"
                    assert (o[n_get] == 0xffffffff).all()                                # nothing written past the end
    finally:
        if three_pass:
            L.mkv_set_tuning(b"pack_three_pass", 0)
        L.mkv_set_tuning(b"pack_threads", PACK_THREADS_DEFAULT)


@pytest.mark.parametrize("scheme,selfd", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_native_engine_one_rank_through_rccl(scheme, selfd, monkeypatch):
    """The C engine (mkv_shard_*: pack, ncclAllToAll, local search+insert, result exchange,
    unpack; two batches in flight) with a one-rank communicator -- the only RCCL shape a
    one-GPU box can run -- against the oracle, results and final table."""
    import numpy as np
    import torch
    from bench import late_imports, sharded_engine as ShardedIndex
    from oracle import oracle as O
    late_imports()
    monkeypatch.setenv("MKV_SHARD_SELF", str(selfd))       # (the engine reads it when it is created) 0: the own slot through ncclAllToAll like the peers'
    mem_p = 16
    rng = np.random.default_rng(77 + scheme)
    n_get, n_set = 6000, 900
    sh = ShardedIndex(mem_p, 1, 0, scheme, "cuda:0", n_get, n_set, comm=None)
    sh.enable_native(None)
    try:
        og = O.geom(mem_p=mem_p, scheme=scheme)
        table = O.new_table(og)
        batches, host = [], []
        for step in range(7):
            ng = n_get if step != 3 else 1234          # a smaller batch in between
            ns = n_set if step != 4 else 0
            keys = rng.integers(1, 40000, size=ng + ns, dtype=np.uint64)
            z = (keys * np.uint64(0x9E3779B97F4A7C15)) ^ (keys >> np.uint64(7))
            sig = (z & np.uint64(0xffffffff)).astype(np.uint32)
            sig[sig == 0] = 1
            hsh = (z >> np.uint64(32)).astype(np.uint32)
            gets = np.stack([sig[:ng], hsh[:ng]], axis=1)
            sets = np.stack([sig[ng:], hsh[ng:], rng.integers(1, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)],
                            axis=1).reshape(-1, 3)
            host.append((gets, sets))
            batches.append(sh.stage(gets, sets))
        sh.run_pipelined(batches)
        torch.cuda.synchronize()
        assert not sh.overflowed()
        for (gets, sets), b in zip(host, batches):
            exp = O.search(og, table, np.ascontiguousarray(gets))
            got = b["out"].cpu().numpy().view(np.uint32)
            assert np.array_equal(got, exp)
            O.insert(og, table, [np.ascontiguousarray(sets)])
        assert np.array_equal(sh.idx.table_numpy(), table)
    finally:
        sh.native.close()


@pytest.mark.parametrize("world,scheme,lanes,selfd,svc,paced,delay_us", [
    # the schedule (local phases on one stream, the rest of a batch on a service stream, three buffer sets,
    # the own slot past the exchange, the peers' slots by grouped send / receive) and its variants
    (4, 0, 3, 1, 2, 1, 0), (8, 0, 3, 1, 2, 1, 0), (2, 1, 3, 1, 1, 1, 0), (8, 0, 2, 1, 1, 1, 0), (4, 0, 4, 1, 3, 1, 0),
    (4, 0, 3, 0, 2, 1, 0), (8, 0, 3, 0, 1, 1, 0), (8, 0, 2, 1, 2, 1, 0), (2, 0, 2, 1, 1, 1, 0), (4, 1, 2, 0, 1, 1, 0),
    # more service streams asked for than buffer sets (the default of several ranks beside MKV_SHARD_LANES=2): clamped
    (4, 0, 2, 1, 3, 1, 0),
    # the un-pack of a batch held back (a slow or skewed exchange) while the stream, not the host, paces the
    # local phases: the local phase of the batch B later writes this rank's own answers while that un-pack
    # has not read its batch's yet -- into the buffer of the other parity
    (4, 0, 3, 1, 2, 0, 150), (8, 0, 2, 1, 1, 0, 150), (2, 1, 2, 1, 1, 0, 300), (4, 0, 3, 1, 2, 1, 150), (8, 0, 3, 0, 2, 0, 150)])
def test_native_engine_schedule_with_several_slots(world, scheme, lanes, selfd, svc, paced, delay_us, monkeypatch):
    """The C engine's W-rank schedule with every peer looped back to this process (two to four
    batches in flight): W slots per batch, a different `where` array for every batch, batches of
    different sizes, results of EVERY batch and the final table against the oracle (canonical
    order of the local table = slot, then batch index).  Wrong event placement between un-pack,
    the local phase and the pack of later batches shows here."""
    import numpy as np
    import torch
    from bench import late_imports, sharded_engine as ShardedIndex
    from megakv_amd.lib import load_library
    from oracle import oracle as O
    late_imports()
    monkeypatch.setenv("MKV_SHARD_SVC", str(svc))              # service streams
    monkeypatch.setenv("MKV_SHARD_LANES", str(lanes))          # batches in flight
    monkeypatch.setenv("MKV_SHARD_SELF", str(selfd))           # the rank's own slot bypasses the exchange
    monkeypatch.setenv("MKV_SHARD_PACED", str(paced))          # 0: the compute stream waits for the requests' event itself
    monkeypatch.setenv("MKV_SHARD_DELAY_UNPACK_US", str(delay_us))
    mem_p = 18
    k = world.bit_length() - 1
    rng = np.random.default_rng(500 + world + 2 * lanes)
    n_get, n_set = 60000, 6000
    # (looped-back peers: a stand-in of the diagnostics build -- the product library refuses the path)
    sh = ShardedIndex(mem_p, world, 0, scheme, "cuda:0", n_get, n_set, comm=None, cap_factor=1.6,
                      lib=load_library(diagnostics=True))
    sh.enable_native(None, loopback=True)
    try:
        og = O.geom(mem_p=mem_p, scheme=scheme, iblock_p=3 - k)
        table = O.new_table(og)
        batches, host = [], []
        for step in range(24):
            ng = n_get if step % 5 != 3 else int(rng.integers(1, n_get))
            ns = n_set if step % 7 != 4 else int(rng.integers(0, 50))
            keys = rng.integers(1, 150000, size=ng + ns, dtype=np.uint64)
            z = (keys * np.uint64(0x9E3779B97F4A7C15)) ^ (keys >> np.uint64(7))
            sig = (z & np.uint64(0xffffffff)).astype(np.uint32)
            sig[sig == 0] = 1
            hsh = (z >> np.uint64(32)).astype(np.uint32)
            gets = np.ascontiguousarray(np.stack([sig[:ng], hsh[:ng]], axis=1))
            sets = np.stack([sig[ng:], hsh[ng:], rng.integers(1, 2**32, size=ns, dtype=np.uint64).astype(np.uint32)],
                            axis=1).reshape(-1, 3)
            host.append((gets, sets))
            batches.append(sh.stage(gets, sets))
        for rep in range(2):                        # the second run re-uses lanes, events and `where` arrays
            for b in batches:
                b["out"].fill_(-1)
            sh.run_pipelined(batches)
            torch.cuda.synchronize()
            assert not sh.overflowed()
            for i, ((gets, sets), b) in enumerate(zip(host, batches)):
                exp = O.search(og, table, gets)
                got = b["out"].cpu().numpy().view(np.uint32)
                assert np.array_equal(got, exp), f"run {rep} batch {i}: results differ"
                shard = sh.shard_of(sets[:, 1]).astype(np.int64)
                O.insert(og, table, [np.ascontiguousarray(sets[shard == s]) for s in range(world)])
            assert np.array_equal(sh.idx.table_numpy(), table)
    finally:
        sh.native.close()


def test_product_library_has_no_looped_back_peers():
    """The looped-back stand-in for the wire (and the wait kernel behind MKV_SHARD_FAKE_XFER_US) are test
    scaffolding: the diagnostics build has them, the product library refuses the path by name."""
    import subprocess
    from bench import late_imports, sharded_engine as ShardedIndex
    from megakv_amd.lib import LIB_PATH
    late_imports()
    sh = ShardedIndex(16, 2, 0, 0, "cuda:0", 1024, 64, comm=None, cap_factor=2.0)
    with pytest.raises(RuntimeError, match="diagnostics build only"):
        sh.enable_native(None, loopback=True)
    names = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "k_wait_us" not in names and "k_loop_group" not in names
