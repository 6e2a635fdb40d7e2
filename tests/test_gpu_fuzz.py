"""Seeded fuzzing of insert / delete / search parity over random geometries
(MEM_P, IBLOCK_P, MAX_CUCKOO_NUM, scheme), fills, key distributions, block
splits and tuning (chunk size, tail rounds, tail work-groups, grid-wide tail rounds).  Every case: GPU result == oracle,
bit for bit, for a sequence of operations on one table.

Two builds of the same sources are under test: the product library, and -- for the cases that FORCE
a path the product takes rarely (tuning "debug_skip": how the tail takes commits back, every eviction
chain left to the tail) -- the diagnostics build, the only one that has that tuning."""
import numpy as np
import pytest
import torch

from megakv_amd import Geometry, HashIndex
from megakv_amd.lib import load_library
from oracle import oracle as O
from tests.gpu_util import dev, host, ogeom, random_table, random_requests, split_blocks, first_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _defaults():
    yield
    for diag in (False, True):
        L = load_library(diag)
        L.mkv_set_tuning(b"tail_rounds", 48)
        L.mkv_set_tuning(b"insert_chunk", 65536)
        L.mkv_set_tuning(b"tail_groups", 32)
        if diag:
            L.mkv_set_tuning(b"grid_min", 2048)
            L.mkv_set_tuning(b"calm_factor", 1)
            L.mkv_set_tuning(b"debug_skip", 0)
        L.mkv_set_tuning(b"round_blocks", 512)
        L.mkv_set_tuning(b"self_help", 1)
        L.mkv_set_tuning(b"merge_round2", 1)
        L.mkv_set_tuning(b"plans", 1)
        L.mkv_set_tuning(b"sorted_min", 0)
        L.mkv_set_tuning(b"sorted_cooldown", 32)


_FIRST = int(__import__("os").environ.get("MKV_FUZZ_FIRST", "0"))       # (campaigns: a later range of seeds)


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(__import__("os").environ.get("MKV_FUZZ_CASES", "48"))))
def test_fuzz_sequence(seed):
    _run(seed, {})


@pytest.mark.parametrize("seed", [194, 429])
@pytest.mark.parametrize("groups,grid_min", [(1, 1 << 30), (8, 0)])
def test_duplicate_evicted_before_its_turn(seed, groups, grid_min):
    """Few distinct signatures: a pending request's duplicate can be evicted by an earlier
    request's chain, after which it needs a chain of its own.  Found by these two seeds once
    the tail ran many rounds (a duplicate used to count as 'cannot turn hard')."""
    _run(seed, {b"tail_rounds": 48, b"tail_groups": groups, b"grid_min": grid_min})


def test_signature_zero_request_turns_hard_behind_others():
    """A request with signature 0 "finds its duplicate" in any empty slot -- which the requests
    before it may fill, after which it needs an eviction chain.  Such a match must not count
    as a duplicate that keeps the request harmless (seed found by the fuzz campaign, chunked
    insert of 13 519 requests with hot keys into a 60 % full table of 32 regions)."""
    _run(2136, {b"tail_groups": 1})
    _run(2136, {b"tail_groups": 32, b"grid_min": 0, b"tail_rounds": 48})


@pytest.mark.parametrize("path", [0, 1, 2])
@pytest.mark.parametrize("chunk", [4096, 65536])
def test_order_violations_of_late_chains(path, chunk):
    """A well-filled table, many inserts: eviction chains that could not run beside the other
    requests walk late and step on buckets later requests were already committed on.  Each of
    the three ways of taking those commits back (a sole commit alone / the commits involved /
    the region suffix) must leave the oracle's table."""
    L = load_library(diagnostics=path != 0)
    if path:
        assert L.mkv_set_tuning(b"debug_skip", path) == 0
    L.mkv_set_tuning(b"insert_chunk", chunk)
    g = Geometry(mem_p=20)
    og = ogeom(g)
    rng = np.random.default_rng(31 + path)
    exp = random_table(g, rng, 0.75, 1 << 31)
    idx = HashIndex(g, lib=L)
    idx.load_table(exp)
    violations = late = 0
    for step in range(4):
        e = random_requests(g, rng, 50000, 1 << 31, hot=0.02 * step)
        blocks = split_blocks(e, g, rng, "region")
        O.insert(og, exp, blocks)
        idx.insert([dev(b) for b in blocks])
        info = idx.last_update_info()
        violations += info["bar_chain"]
        late += info["bar_class"]
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (path, chunk, step, first_diff(got, exp), info)
    assert late > 0 and violations > 0, (late, violations)     # the paths in question were taken


@pytest.mark.parametrize("blocks", [1024, 4096])
def test_rounds_on_more_work_groups_than_list_places(blocks):
    """Rounds 1 and 2 list what they leave to the tail per work-group, 512 places; on more work-groups
    (tuning round_blocks) several share a place and ADD to its count -- which a kernel before them
    must have cleared.  Several passes in a row on a well-filled table (every pass leaves the tail
    requests), whole table against the oracle after each."""
    L = load_library()
    L.mkv_set_tuning(b"round_blocks", blocks)
    g = Geometry(mem_p=20)
    og = ogeom(g)
    rng = np.random.default_rng(4242 + blocks)
    exp = random_table(g, rng, 0.7, 1 << 31)
    idx = HashIndex(g)
    idx.load_table(exp)
    listed = 0
    for step in range(5):
        e = random_requests(g, rng, 60000, 1 << 31, hot=0.05 * step)
        blocks_ = split_blocks(e, g, rng, "region")
        O.insert(og, exp, blocks_)
        idx.insert([dev(b) for b in blocks_])
        info = idx.last_update_info()
        listed += info["tail_parallel"] + info["tail_serial"]
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (blocks, step, first_diff(got, exp), info)
    assert listed > 0                                    # the tail did have work


DIAG_ONLY = (b"grid_min", b"grid_progress", b"calm_factor", b"mark_alone_blocks", b"debug_skip", b"debug_first_stamp")


def _run(seed, forced):
    rng = np.random.default_rng(1000 + seed)
    env = __import__("os").environ
    # how the tail deals with an order violation of a late eviction chain: as designed (a sole later
    # commit is taken back alone), always the general way, always the region-suffix fall-back
    # (+ 4: rounds 1 and 2 leave every eviction chain to the tail) -- anything but "as designed"
    # is the diagnostics build's
    rng3 = np.random.default_rng(11000 + seed)
    skip = int(rng3.choice([0, 0, 1, 2])) + int(rng3.choice([0, 0, 0, 4]))
    skip = int(env.get("MKV_FUZZ_DEBUG_SKIP", forced.get(b"debug_skip", skip)))
    # The knobs of tests and probes (DIAG_ONLY) exist in the diagnostics build only -- the same sources; the
    # product library has their defaults as constants.  A case that forces one of them, and every other seed,
    # runs on the diagnostics build with all knobs drawn; the rest on the product library with its own sixteen.
    diag = skip != 0 or seed % 2 == 1 or any(k in DIAG_ONLY for k in forced) or \
        any(("MKV_FUZZ_" + k.decode().upper()) in env for k in DIAG_ONLY)
    L = load_library(diagnostics=diag)
    big = int(env.get("MKV_FUZZ_BIG", "0"))          # 1: tables up to 2 MiB, batches up to 300 K requests
    mem_p = int(rng.integers(10, 19)) + (int(rng.integers(0, 4)) if big else 0)
    iblock_p = int(rng.integers(0, min(5, mem_p - 7) + 1))
    scheme = int(rng.integers(0, 2))
    max_cuckoo = int(rng.choice([0, 1, 2, 5, 6]))
    g = Geometry(mem_p=mem_p, scheme=scheme, iblock_p=iblock_p, max_cuckoo=max_cuckoo)
    og = ogeom(g)
    L.mkv_set_tuning(b"tail_rounds", int(rng.choice([0, 1, 3, 48])))
    L.mkv_set_tuning(b"insert_chunk", int(rng.choice([1024, 2048, 8192, 65536])))
    rng2 = np.random.default_rng(5000 + seed)        # tail shape: work-groups, grid-wide rounds, calm-mode chunks
    L.mkv_set_tuning(b"tail_groups", int(rng2.choice([1, 2, 5, 32])))
    gm, cf = int(rng2.choice([0, 0, 64, 2048])), int(rng2.choice([1, 4]))
    if diag:
        assert L.mkv_set_tuning(b"grid_min", gm) == 0 and L.mkv_set_tuning(b"calm_factor", cf) == 0
    if skip:
        assert L.mkv_set_tuning(b"debug_skip", skip) == 0
    # work-groups of rounds 1 and 2: few make a work-group's queue of held-back requests run over
    L.mkv_set_tuning(b"round_blocks", int(rng3.choice([1, 3, 64, 512, 512])))
    if seed % 7 == 3:
        L.mkv_set_tuning(b"round_blocks", 2048)      # (more work-groups than places in the rounds' lists for the tail)
    # round 1's self-help of a bucket's second registrant, round 2 inside the tail launch: on (the default) or off
    rng4 = np.random.default_rng(17000 + seed)
    L.mkv_set_tuning(b"self_help", int(rng4.choice([1, 1, 0])))
    L.mkv_set_tuning(b"merge_round2", int(rng4.choice([1, 1, 0])))
    # round 1 carrying out the mark pass's plans (round 5): on (the default) or off
    L.mkv_set_tuning(b"plans", int(np.random.default_rng(19000 + seed).choice([1, 1, 1, 0])))
    # big insert passes (round 6: partition into bins, apply per bucket -- mkv_upd_apply.h): off (the default), or forced
    # at every size for one seed in four (never backing off)
    L.mkv_set_tuning(b"sorted_min", int(np.random.default_rng(23000 + seed).choice([0, 0, 0, 1])))
    L.mkv_set_tuning(b"sorted_cooldown", 0)
    for name in (b"tail_groups", b"grid_min", b"calm_factor", b"tail_rounds", b"insert_chunk", b"round_blocks",
                 b"self_help", b"merge_round2", b"plans", b"sorted_min", b"sorted_cooldown"):   # bisecting aid
        v = __import__("os").environ.get("MKV_FUZZ_" + name.decode().upper())
        if v is not None:
            assert L.mkv_set_tuning(name, int(v)) == 0, name
    for name, v in forced.items():
        if name != b"debug_skip":
            assert L.mkv_set_tuning(name, int(v)) == 0, name
    fill = float(rng.choice([0.0, 0.3, 0.6, 0.85, 0.97]))
    sig_space = int(rng.choice([40, 5000, 1 << 20, 1 << 31]))
    exp = random_table(g, rng, fill, sig_space)
    idx = HashIndex(g, lib=L)
    idx.load_table(exp)
    nslots = g.nbuckets * 8
    # request shapes beside uniformly random ones, drawn apart from the main stream of random
    # numbers so that the sequences of earlier rounds stay what they were
    shape = str(np.random.default_rng(7000 + seed).choice(["random", "random", "random", "ref", "cluster"]))
    for step in range(3):
        op = rng.choice(["insert", "insert", "delete", "search"])
        n = int(rng.integers(1, max(2, min(300000 if big else 40000, nslots))))
        e = random_requests(g, rng, n, sig_space, hot=float(rng.choice([0.0, 0.1, 0.6])),
                            zero_low=float(rng.choice([0.0, 0.1])), zeros=float(rng.choice([0.0, 0.03])),
                            shape=shape)
        ctx = dict(seed=seed, mem_p=mem_p, iblock_p=iblock_p, scheme=scheme, max_cuckoo=max_cuckoo, fill=fill,
                   step=step, op=str(op), n=n)
        if op == "insert":
            how = str(rng.choice(["single", "region", "app", "uneven"]))
            if iblock_p == 0 and how == "app":
                how = "single"
            blocks = split_blocks(e, g, rng, how)
            O.insert(og, exp, blocks)
            idx.insert([dev(b.reshape(-1, 3)) for b in blocks],
                       hint=None if rng.random() < 0.7 else 0)      # hinted (chunked) or plain single pass
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (ctx, how, first_diff(got, exp), idx.last_update_info())
        elif op == "delete":
            # make half of the deletes hit stored entries
            occ = np.argwhere(exp[:, :8] != 0)
            if len(occ):
                pick = occ[rng.integers(0, len(occ), size=n // 2)]
                e[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
                e[: len(pick), 1] = pick[:, 0]
                e[: len(pick), 2] = exp[pick[:, 0], 8 + pick[:, 1]]
            O.delete(og, exp, e)
            idx.delete(dev(e))
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (ctx, first_diff(got, exp))
        else:
            q = np.ascontiguousarray(e[:, :2])
            out = host(idx.search(dev(q), zero_miss=bool(rng.integers(0, 2))))
            ref = O.search(og, exp, q)
            assert np.array_equal(out, ref), ctx


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MKV_FUZZ_CYCLE_CASES", "24"))))
def test_fuzz_cycles(seed):
    """mkv_cycle (search + delete + insert of one worker in one call, the insert's mark pass
    inside the search launch: the path bench.py measures) over random geometries, tables,
    request mixes and launch variants, several cycles in a row against the oracle."""
    _run_cycles(seed, deletes=True)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MKV_FUZZ_NODEL_CASES", "24"))))
def test_fuzz_cycles_without_deletes(seed):
    """The same without deletes, always fused, one pass -- the shape of bench.py's step -- on 1, 3, 40
    or 128 mark work-groups."""
    mb = int(np.random.default_rng(13000 + seed).choice([1, 3, 40, 128]))
    for d in (False, True):
        load_library(d).mkv_set_tuning(b"mark_blocks", mb)
    try:
        _run_cycles(seed, deletes=False)
    finally:
        for d in (False, True):
            load_library(d).mkv_set_tuning(b"mark_blocks", 208)


def _run_cycles(seed, deletes):
    rng = np.random.default_rng(9000 + seed)
    diag = seed % 2 == 1                             # (every other seed on the diagnostics build, whose tail has all its knobs)
    L = load_library(diagnostics=diag)
    mem_p = int(rng.integers(10, 19))
    iblock_p = int(rng.integers(0, min(5, mem_p - 7) + 1))
    scheme = int(rng.integers(0, 2))
    g = Geometry(mem_p=mem_p, scheme=scheme, iblock_p=iblock_p, max_cuckoo=int(rng.choice([1, 2, 5])))
    og = ogeom(g)
    L.mkv_set_tuning(b"tail_rounds", int(rng.choice([0, 3, 48])))
    L.mkv_set_tuning(b"insert_chunk", int(rng.choice([1024, 8192, 65536])))
    L.mkv_set_tuning(b"tail_groups", int(rng.choice([1, 5, 32])))
    gm = int(rng.choice([0, 64, 2048]))
    if diag:
        assert L.mkv_set_tuning(b"grid_min", gm) == 0
    rng4 = np.random.default_rng(17000 + seed)
    L.mkv_set_tuning(b"self_help", int(rng4.choice([1, 1, 0])))
    L.mkv_set_tuning(b"merge_round2", int(rng4.choice([1, 1, 0])))
    L.mkv_set_tuning(b"plans", int(np.random.default_rng(19000 + seed).choice([1, 1, 1, 0])))
    if not deletes:
        L.mkv_set_tuning(b"insert_chunk", 65536)     # (one pass)
    # big insert passes (round 6): the partition rides in the cycle's search launch -- forced for one seed in four
    sm = __import__("os").environ.get("MKV_FUZZ_SORTED_MIN")
    L.mkv_set_tuning(b"sorted_min", int(sm) if sm is not None else int(np.random.default_rng(23000 + seed).choice([0, 0, 0, 1])))
    L.mkv_set_tuning(b"sorted_cooldown", 0)
    sig_space = int(rng.choice([40, 5000, 1 << 20]))
    exp = random_table(g, rng, float(rng.choice([0.0, 0.3, 0.7, 0.9])), sig_space)
    idx = HashIndex(g, lib=L)
    idx.load_table(exp)
    nslots = g.nbuckets * 8
    memset = bool(rng.integers(0, 2))
    overlap = str(rng.choice(["fused", "fused", "side", "none"]))
    if not deletes:
        memset, overlap = False, "fused"
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for step in range(3):
            nq = int(rng.integers(0, 20000))
            nd = int(rng.integers(0, 3000))
            if not deletes:
                nq, nd = max(nq, 1), 0
            ni = int(rng.integers(0, max(2, min(30000, nslots))))
            hot = float(rng.choice([0.0, 0.1, 0.5]))
            q = random_requests(g, rng, nq, sig_space, hot=hot, zeros=float(rng.choice([0.0, 0.03])))[:, :2].copy()
            d = random_requests(g, rng, nd, sig_space)
            e = random_requests(g, rng, ni, sig_space, hot=hot, zero_low=float(rng.choice([0.0, 0.1])),
                                zeros=float(rng.choice([0.0, 0.03])))
            occ = np.argwhere(exp[:, :8] != 0)
            if len(occ):
                pick = occ[rng.integers(0, len(occ), size=min(nq, nd) // 2)]
                q[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
                q[: len(pick), 1] = pick[:, 0]
                d[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
                d[: len(pick), 1] = pick[:, 0]
                d[: len(pick), 2] = exp[pick[:, 0], 8 + pick[:, 1]]
            how = str(rng.choice(["single", "region", "uneven"]))
            blocks = split_blocks(e, g, rng, how)
            dblocks = [dev(b.reshape(-1, 3)) for b in blocks]
            ptrs, cnts, keep = idx.stage_blocks(dblocks)
            out = dev(np.full((max(nq, 1), 2), 0x5555AAAA, np.uint32))
            dq, dd = (dev(q) if nq else None), (dev(d) if nd else None)
            idx.cycle(dq, out if nq else None, ptrs, cnts, len(keep), delem=dd, memset=memset, overlap=overlap,
                      stream=stream, insert_hint=int(rng.choice([0, ni])))
            exp_out = O.search(og, exp, q)
            if nd:
                O.delete(og, exp, d)
            O.insert(og, exp, blocks)
            stream.synchronize()
            ctx = dict(seed=seed, step=step, mem_p=mem_p, iblock_p=iblock_p, scheme=scheme, nq=nq, nd=nd, ni=ni,
                       memset=memset, overlap=overlap, how=how)
            if nq:
                assert np.array_equal(host(out)[:nq], exp_out), ctx
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (ctx, first_diff(got, exp))


@pytest.mark.parametrize("self_help", [1, 0])
@pytest.mark.parametrize("plans", [1, 0])
@pytest.mark.parametrize("j_first", [False, True])
def test_chain_hint_between_owner_and_self_helping_second_registrant(self_help, plans, j_first):
    """The advisor's round-4 case.  Requests c < h < j of one pass: c and j have bucket P as their primary bucket
    (different signatures, P has room); h finds its own two buckets full and its eviction chain's first victim
    moves on to P -- the mark pass's walk ahead registers h there, without a returning atomic, so second[P] names j
    and j helps itself in round 1 beside c, AHEAD of the pending h, whose chain writes P later.  The tail finds
    j's commit through second[P] when the chain walks and takes it back: the table must be the oracle's.  Run on
    the diagnostics build with and without the self-help, with plans on and off, and with the roles of the later
    requests swapped (j_first: the second registrant precedes the chain, no violation)."""
    L = load_library(diagnostics=True)
    L.mkv_set_tuning(b"self_help", self_help)
    L.mkv_set_tuning(b"plans", plans)
    g = Geometry(mem_p=12, iblock_p=0, max_cuckoo=5)              # 64 buckets, one region
    og = ogeom(g)
    mask = g.hash_mask
    rng = np.random.default_rng(5)
    table = np.zeros((g.nbuckets, 16), np.uint32)
    P, Ph = 9, 20
    h_sig, h_hash = 0x5a5a0003, Ph                                 # victim slot 3 (sig & 7)
    Ah = (h_hash ^ h_sig) & mask
    assert Ah not in (P, Ph)
    for b in (Ph, Ah):                                              # h's two buckets: full, no duplicate of h
        table[b, :8] = rng.integers(1 << 20, 1 << 30, size=8)
        table[b, 8:] = rng.integers(1, 1 << 30, size=8)
    victim = ((h_hash ^ P) & mask) | 0x77000000                    # ... and the victim's next bucket is P
    assert (h_hash ^ victim) & mask == P and victim & 7 != 3 or True
    table[Ah, 3] = victim
    table[P, :3] = [111, 222, 333]                                  # P: three entries, room for more
    table[P, 8:11] = [1, 2, 3]
    c = [0x10000001, P, 1001]
    h = [h_sig, h_hash, 2002]
    j = [0x20000002, P | 0x40000000, 3003]                          # (other hash bits above the bucket index)
    reqs = np.array([c, j, h] if j_first else [c, h, j], np.uint32)
    exp = table.copy()
    O.insert(og, exp, [reqs])
    idx = HashIndex(g, lib=L)
    idx.load_table(table)
    idx.insert([dev(reqs)])
    got = idx.table_numpy()
    info = idx.last_update_info()
    L.mkv_set_tuning(b"self_help", 1)
    L.mkv_set_tuning(b"plans", 1)
    assert np.array_equal(got, exp), (first_diff(got, exp), info)
    # the chain did go through P (the case is the one it claims to be)
    assert victim in exp[P, :8].tolist()
