"""Long sequences on ONE table: hundreds of consecutive scheduler cycles (search + delete + insert through
`mkv_cycle`, the fused launch, with plain three-call cycles mixed in) while the load factor swings from empty to
nearly full and back.  The fuzz cases are three operations long; what only shows over many passes -- the
pass's epilogue left pending again and again (Ctl::ep_pending), the host-mapped flag words switching the
insert path's modes on and off (round 2 inside the tail launch or not, the walks ahead flagging or not,
chunked passes after late chains), round stamps running on -- needs a long run.  The whole table and every
search result are compared with the oracle after every cycle."""
import numpy as np
import pytest
import torch

from megakv_amd import Geometry, HashIndex
from megakv_amd.lib import load_library
from oracle import oracle as O
from tests.gpu_util import dev, host, ogeom, random_requests, split_blocks, first_diff

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scheme,diag,mem_p,scale,steps", [(0, False, 17, 1, 360), (0, True, 17, 1, 360), (1, False, 17, 1, 360),
                                                            (0, False, 21, 16, 180), (1, True, 21, 16, 120)])
def test_hundreds_of_cycles_on_one_table(scheme, diag, mem_p, scale, steps):
    L = load_library(diagnostics=diag)
    # (mem_p 17: 2 048 buckets, 16 384 slots, four regions, batches of hundreds; 21: sixteen times that)
    g = Geometry(mem_p=mem_p, scheme=scheme, iblock_p=2, max_cuckoo=5)
    og = ogeom(g)
    rng = np.random.default_rng(777 + scheme + 10 * diag + mem_p)
    exp = np.zeros((g.nbuckets, 16), np.uint32)
    idx = HashIndex(g, lib=L)
    idx.load_table(exp)
    nslots = g.nbuckets * 8
    sig_space = 1 << 20
    stream = torch.cuda.Stream()
    modes = set()
    with torch.cuda.stream(stream):
        for step in range(steps):
            # the load factor the phase aims at: up for 120 cycles, down for 60, up again, ...
            filling = (step // 60) % 3 != 2
            lf = np.count_nonzero(exp[:, :8]) / nslots
            ni = int(rng.integers(0, (700 if filling else 120) * scale))
            nd = int(rng.integers(0, (80 if filling else 900) * scale))
            nq = int(rng.integers(1, 1500 * scale))
            hot = float(rng.choice([0.0, 0.0, 0.2]))
            q = random_requests(g, rng, nq, sig_space, hot=hot)[:, :2].copy()
            d = random_requests(g, rng, nd, sig_space)
            e = random_requests(g, rng, ni, sig_space, hot=hot, zeros=float(rng.choice([0.0, 0.01])))
            occ = np.argwhere(exp[:, :8] != 0)
            if len(occ):
                pick = occ[rng.integers(0, len(occ), size=nq // 2)]
                q[: len(pick), 0] = exp[pick[:, 0], pick[:, 1]]
                q[: len(pick), 1] = pick[:, 0]
                pick = occ[rng.integers(0, len(occ), size=nd)]
                d[:, 0] = exp[pick[:, 0], pick[:, 1]]
                d[:, 1] = pick[:, 0]
                d[:, 2] = exp[pick[:, 0], 8 + pick[:, 1]]
            blocks = split_blocks(e, g, rng, str(rng.choice(["single", "region", "uneven"])))
            dblocks = [dev(b.reshape(-1, 3)) for b in blocks]
            ptrs, cnts, keep = idx.stage_blocks(dblocks)
            out = dev(np.full((nq, 2), 0x5555AAAA, np.uint32))
            three_call = step % 7 == 3
            if three_call:
                out.zero_()
                idx.search(dev(q), out=out, zero_miss=False, stream=stream)
                if nd:
                    idx.delete(dev(d), stream=stream)
                if ni:
                    idx.insert_staged(ptrs, cnts, len(keep), stream=stream, hint=0)
            else:
                idx.cycle(dev(q), out, ptrs, cnts, len(keep), delem=dev(d) if nd else None, memset=False, overlap="fused",
                          stream=stream, insert_hint=int(rng.choice([0, ni])))
            exp_out = O.search(og, exp, q)
            if nd:
                O.delete(og, exp, d)
            O.insert(og, exp, blocks)
            stream.synchronize()
            ctx = dict(step=step, lf=round(lf, 3), nq=nq, nd=nd, ni=ni, three_call=three_call, scheme=scheme, diag=diag, mem_p=mem_p)
            assert np.array_equal(host(out)[:nq], exp_out), ctx
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (ctx, first_diff(got, exp))
            if ni and not three_call:
                info = idx.last_update_info(stream, cycle_insert=True)
                modes.add((info["tail_serial"] + info["tail_parallel"] > 0, info["chains"] > 0, info["bar_class"] > 0))
    # the run saw passes with and without work for the tail, and -- the cuckoo scheme, near full -- eviction chains
    assert len(modes) >= 2, modes
    if scheme == 0:
        assert any(m[1] or m[2] for m in modes), modes


@pytest.mark.parametrize("scheme", [0, 1])
def test_big_passes_wide_round_1_and_its_fall_back(scheme):
    """Passes announced with more than 131 072 requests: round 1 runs on a lane per request (2 048 work-groups, counts and
    lists through shared places) once the pass before was mostly plan-carrying, and keeps to 512 work-groups when it
    was not (primary buckets full: more than one request in 128 without a plan) -- the verdict is left one pass
    ahead (diagnostics build: mkv_debug_wide_round).  A table filled from empty to past load factor 0.6 in passes of
    150 000 new keys, then thinned out by deletes and filled again: the verdict goes 1 -> 0 -> 1, every table exact."""
    L = load_library(diagnostics=True)
    # (the product's defaults, whatever the tests before left: one pass per call, 512 work-groups for the rounds, plans on)
    for name, v in ((b"insert_chunk", 524288), (b"round_blocks", 512), (b"plans", 1), (b"tail_rounds", 48), (b"tail_groups", 32)):
        assert L.mkv_set_tuning(name, v) == 0
    try:
        _big_passes(L, scheme)
    finally:
        L.mkv_set_tuning(b"insert_chunk", 65536)          # (what the other suites' fixtures leave)


def _big_passes(L, scheme):
    g = Geometry(mem_p=25, scheme=scheme, iblock_p=3, max_cuckoo=5)      # 2^19 buckets, 4 M slots
    og = ogeom(g)
    rng = np.random.default_rng(99 + scheme)
    exp = np.zeros((g.nbuckets, 16), np.uint32)
    idx = HashIndex(g, lib=L)
    idx.load_table(exp)
    nslots = g.nbuckets * 8
    verdicts = []

    def insert_pass(n):
        e = random_requests(g, rng, n, 1 << 31)
        blocks = split_blocks(e, g, rng, "region")
        O.insert(og, exp, blocks)
        idx.insert([dev(b) for b in blocks])              # (hint = the batch's size)
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (scheme, len(verdicts), first_diff(got, exp), idx.last_update_info())
        verdicts.append((round(np.count_nonzero(exp[:, :8]) / nslots, 2), L.mkv_debug_wide_round(0)))

    for _ in range(18):                                    # 18 x 150 000 keys: load factor 0 -> ~0.6
        insert_pass(150000)
    occ = np.argwhere(exp[:, :8] != 0)
    pick = occ[rng.permutation(len(occ))[: int(len(occ) * 0.8)]]
    d = np.stack([exp[pick[:, 0], pick[:, 1]], pick[:, 0].astype(np.uint32), exp[pick[:, 0], 8 + pick[:, 1]]], axis=1).astype(np.uint32)
    O.delete(og, exp, d)
    idx.delete(dev(d))
    assert np.array_equal(idx.table_numpy(), exp)
    for _ in range(4):
        insert_pass(150000)
    v = [w for _, w in verdicts]
    assert v[1] == 1 and 0 in v[4:18] and v[-1] == 1, verdicts      # calm at first, crowded later, calm again once thinned out
