"""GPU parity: the HIP kernels (through the C ABI) against the CPU oracle on
identical inputs, bit for bit -- search output words and full table images."""
import numpy as np
import pytest
import torch

from megakv_amd import Geometry, HashIndex, SCHEME_CUCKOO, SCHEME_2CHOICE
from megakv_amd.lib import load_library
from oracle import oracle as O
from tests.gpu_util import (dev, host, ogeom, random_table, random_requests, split_blocks,
                            run_insert_case, first_diff)

pytestmark = pytest.mark.gpu

SCHEMES = [SCHEME_CUCKOO, SCHEME_2CHOICE]


def set_tuning(name, value, diagnostics=False):
    assert load_library(diagnostics).mkv_set_tuning(name.encode(), value) == 0


@pytest.fixture(autouse=True)
def _defaults():
    yield
    set_tuning("tail_rounds", 48)
    set_tuning("insert_chunk", 65536)
    set_tuning("search_u", 4)
    set_tuning("search_nt", 0)
    set_tuning("search_u", 4, True)
    set_tuning("search_wide", 0, True)
    set_tuning("search_lds", 0, True)


def test_product_library_has_no_diagnostics():
    """The library a host links exports no mkv_debug_* entry point and refuses the tunings that
    exist for tests and measurements of withdrawn layouts; the diagnostics build has them."""
    L, D = load_library(), load_library(diagnostics=True)
    for name in ("mkv_debug_request", "mkv_debug_tail_phases", "mkv_debug_stamp_resets", "mkv_debug_random_read"):
        assert not hasattr(L, name) and hasattr(D, name)
    for name, dflt in ((b"debug_skip", 0), (b"debug_first_stamp", 0), (b"search_wide", 0), (b"search_lds", 0),
                       (b"pack_three_pass", 0), (b"grid_min", 2048), (b"grid_progress", 96), (b"calm_factor", 1),
                       (b"mark_alone_blocks", 1024)):
        assert L.mkv_set_tuning(name, dflt) != 0
        assert D.mkv_set_tuning(name, dflt) == 0


# ---------------------------------------------------------------- search ----
@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 63, 64, 65, 255, 1000, 65539])
@pytest.mark.parametrize("mem_p", [12, 16, 22])
def test_search_matches_oracle_ragged(n, mem_p):
    g = Geometry(mem_p=mem_p)
    rng = np.random.default_rng(1000 + n + mem_p)
    img = random_table(g, rng, 0.7, sig_space=40)      # few signatures: hits and in-bucket duplicates
    q = random_requests(g, rng, n, 40)[:, :2].copy()
    idx = HashIndex(g)
    idx.load_table(img)
    for zero_miss in (True, False):
        exp = np.zeros((n, 2), np.uint32) if zero_miss else np.full((n, 2), 0xABCD1234, np.uint32)
        O.search(ogeom(g), img, q, out=exp)
        out = dev(np.full((max(n, 1), 2), 0xABCD1234, np.uint32))[:n]
        if n:
            idx.search(dev(q), out=out, zero_miss=zero_miss)
        torch.cuda.synchronize()
        assert np.array_equal(host(out), exp), f"zero_miss={zero_miss}"


@pytest.mark.parametrize("u", [1, 2, 4, 8])
@pytest.mark.parametrize("nt", [0, 1])
def test_search_tuning_variants_agree(u, nt):
    set_tuning("search_u", u)
    set_tuning("search_nt", nt)
    g = Geometry(mem_p=20)
    rng = np.random.default_rng(77)
    img = random_table(g, rng, 0.5, sig_space=1000)
    q = random_requests(g, rng, 30011, 1000)[:, :2].copy()
    idx = HashIndex(g)
    idx.load_table(img)
    out = idx.search(dev(q))
    assert np.array_equal(host(out), O.search(ogeom(g), img, q))


@pytest.mark.parametrize("u", [1, 4])
@pytest.mark.parametrize("zero_miss", [True, False])
@pytest.mark.parametrize("n", [1, 9, 64, 30011])
def test_search_wide_layout_agrees(u, zero_miss, n):
    """The 8-lanes-per-request layout (one 64-byte access per bucket; slower, kept as a
    measured alternative: DESIGN 5.1) gives the same results, ragged tails included."""
    set_tuning("search_u", u, True)
    set_tuning("search_wide", 1, True)
    g = Geometry(mem_p=18)
    rng = np.random.default_rng(78 + n)
    img = random_table(g, rng, 0.6, sig_space=60)          # few signatures: in-bucket duplicates
    q = random_requests(g, rng, n, 60)[:, :2].copy()
    idx = HashIndex(g, lib=load_library(diagnostics=True))
    idx.load_table(img)
    exp = np.zeros((n, 2), np.uint32) if zero_miss else np.full((n, 2), 0xABCD1234, np.uint32)
    O.search(ogeom(g), img, q, out=exp)
    out = dev(np.full((n, 2), 0xABCD1234, np.uint32))
    idx.search(dev(q), out=out, zero_miss=zero_miss)
    torch.cuda.synchronize()
    assert np.array_equal(host(out), exp)


@pytest.mark.parametrize("zero_miss", [True, False])
@pytest.mark.parametrize("n", [1, 9, 64, 65, 30011])
def test_search_lds_ballot_layout_agrees(zero_miss, n):
    """The reference's structure on 64 lanes -- signature rows staged in LDS, a wave ballot per
    request, ffs/clz for the slot, a dependent load for the location (the search the task's
    north star sketches; measured and kept as an alternative: DESIGN 5.1) -- gives the same
    results, in-bucket duplicates and ragged tails included."""
    set_tuning("search_lds", 1, True)
    g = Geometry(mem_p=18)
    rng = np.random.default_rng(178 + n)
    img = random_table(g, rng, 0.6, sig_space=60)
    q = random_requests(g, rng, n, 60, zero_low=0.1)[:, :2].copy()
    idx = HashIndex(g, lib=load_library(diagnostics=True))
    idx.load_table(img)
    exp = np.zeros((n, 2), np.uint32) if zero_miss else np.full((n, 2), 0xABCD1234, np.uint32)
    O.search(ogeom(g), img, q, out=exp)
    out = dev(np.full((n, 2), 0xABCD1234, np.uint32))
    idx.search(dev(q), out=out, zero_miss=zero_miss)
    torch.cuda.synchronize()
    assert np.array_equal(host(out), exp)


def test_search_kat_image_reference_test():
    """libgpuhash/test/back/py_search_stream.c:105-129: every query hits in both
    buckets with loc 1."""
    g = Geometry(mem_p=24)
    img = np.empty((g.nbuckets, 16), np.uint32)
    img[:, :8] = np.arange(1, 9, dtype=np.uint32)
    img[:, 8:] = 1
    rng = np.random.default_rng(5)
    n = 1 << 18
    q = np.stack([rng.integers(1, 9, size=n), rng.integers(0, g.nbuckets, size=n)], axis=1).astype(np.uint32)
    idx = HashIndex(g)
    idx.load_table(img)
    out = idx.search(dev(q), zero_miss=False)
    assert (host(out) == 1).all()


# ---------------------------------------------------------------- delete ----
@pytest.mark.parametrize("mem_p,n", [(12, 500), (14, 3000), (18, 20000), (20, 1)])
def test_delete_matches_oracle(mem_p, n):
    g = Geometry(mem_p=mem_p)
    rng = np.random.default_rng(31 + n)
    img = random_table(g, rng, 0.8, sig_space=30)
    # small loc space so (sig, loc) pairs really match, with repeats in the batch
    img[:, 8:] = np.where(img[:, :8] != 0, rng.integers(1, 4, size=(g.nbuckets, 8)), 0)
    d = random_requests(g, rng, n, 30, hot=0.2)
    d[:, 2] = rng.integers(1, 4, size=n)
    exp = img.copy()
    O.delete(ogeom(g), exp, d)
    idx = HashIndex(g)
    idx.load_table(img)
    idx.delete(dev(d))
    got = idx.table_numpy()
    assert np.array_equal(got, exp), first_diff(got, exp)


def test_delete_kat_image_reference_test():
    """libgpuhash/test/back/py_delete_stream.c:100-117"""
    g = Geometry(mem_p=20)
    img = np.zeros((g.nbuckets, 16), np.uint32)
    img[:, :8] = np.arange(1, 9, dtype=np.uint32)
    rng = np.random.default_rng(6)
    n = 4096
    d = np.stack([rng.integers(1, 9, size=n), rng.integers(0, g.nbuckets, size=n),
                  np.zeros(n)], axis=1).astype(np.uint32)
    exp = img.copy()
    O.delete(ogeom(g), exp, d)
    idx = HashIndex(g)
    idx.load_table(img)
    idx.delete(dev(d))
    assert np.array_equal(idx.table_numpy(), exp)


# ---------------------------------------------------------------- insert ----
CASES = [
    # mem_p, n, fill, sig_space, hot, zero_low, zeros, blocks
    (12, 1, 0.0, 1000, 0.0, 0.0, 0.0, "single"),
    (12, 7, 0.5, 1000, 0.0, 0.0, 0.0, "region"),
    (12, 300, 0.0, 50, 0.0, 0.0, 0.0, "region"),        # 64 buckets: heavy sharing, dup sigs
    (12, 2000, 0.6, 100000, 0.1, 0.05, 0.02, "app"),     # overflow + eviction chains everywhere
    (14, 5000, 0.9, 100000, 0.0, 0.0, 0.0, "uneven"),    # nearly full: chains, dropped victims
    (16, 20000, 0.3, 1 << 30, 0.3, 0.0, 0.0, "region"),  # hot duplicate keys
    (16, 20000, 0.5, 1 << 30, 0.0, 0.2, 0.01, "app"),    # alternate == primary cases
    (20, 60000, 0.2, 1 << 31, 0.0, 0.0, 0.0, "region"),  # the reference test's regime
    (20, 60000, 0.85, 1 << 31, 0.05, 0.0, 0.0, "single"),
    (22, 200000, 0.8, 1 << 31, 0.0, 0.0, 0.0, "region"),  # many eviction chains, few conflicts between them
    (22, 200000, 0.6, 1 << 31, 0.02, 0.01, 0.0, "app"),
]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"p{c[0]}-n{c[1]}-f{c[2]}-{c[7]}")
def test_insert_matches_oracle(case, scheme):
    mem_p, n, fill, sig_space, hot, zero_low, zeros, how = case
    g = Geometry(mem_p=mem_p, scheme=scheme)
    rng = np.random.default_rng(hash((mem_p, n, scheme)) & 0xFFFF)
    img = random_table(g, rng, fill, min(sig_space, 2**31))
    e = random_requests(g, rng, n, sig_space, hot, zero_low, zeros)
    blocks = split_blocks(e, g, rng, how)
    got, exp, info = run_insert_case(g, img, blocks)
    assert np.array_equal(got, exp), (first_diff(got, exp), info)
    done = info["round1"] + info["round2"] + info["tail_parallel"] + info["tail_serial"] - info["undone"]
    assert done <= info["total"]


@pytest.mark.parametrize("rounds", [0, 1, 5])
@pytest.mark.parametrize("scheme", SCHEMES)
def test_insert_result_independent_of_round_count(rounds, scheme):
    set_tuning("tail_rounds", rounds)
    g = Geometry(mem_p=14, scheme=scheme)
    rng = np.random.default_rng(rounds)
    img = random_table(g, rng, 0.7, 1 << 20)
    e = random_requests(g, rng, 6000, 1 << 20, hot=0.2, zero_low=0.05)
    got, exp, info = run_insert_case(g, img, split_blocks(e, g, rng, "region"))
    assert np.array_equal(got, exp), (first_diff(got, exp), info)


def test_eviction_chains_commit_in_parallel():
    """At LF ~0.8 thousands of requests are hard; chains whose buckets nobody else
    touches commit inside the grid-wide rounds (info['chains']), the rest behind
    the region barrier -- the table still equals the oracle's."""
    g = Geometry(mem_p=24)
    rng = np.random.default_rng(21)
    img = random_table(g, rng, 0.8, 1 << 31)
    e = random_requests(g, rng, 100000, 1 << 31)
    got, exp, info = run_insert_case(g, img, split_blocks(e, g, rng, "region"))
    assert np.array_equal(got, exp), (first_diff(got, exp), info)
    assert info["chains"] > 100, info
    assert info["first_hard"] != 0xFFFFFFFF


@pytest.mark.parametrize("chunk", [1024, 4096, 20000])
@pytest.mark.parametrize("scheme", SCHEMES)
def test_chunked_insert_equals_single_pass(chunk, scheme):
    """A size hint makes the library run the batch in canonical-index chunks; the
    table must not depend on the chunk size (also through mkv_cycle)."""
    set_tuning("insert_chunk", chunk)
    g = Geometry(mem_p=18, scheme=scheme)
    rng = np.random.default_rng(chunk)
    img = random_table(g, rng, 0.75, 1 << 31)
    e = random_requests(g, rng, 30000, 1 << 31, hot=0.05, zero_low=0.02)
    blocks = split_blocks(e, g, rng, "uneven")
    got, exp, info = run_insert_case(g, img, blocks)
    assert np.array_equal(got, exp), (first_diff(got, exp), info)
    assert info["total"] == 30000
    # and through the fused cycle entry
    idx = HashIndex(g)
    idx.load_table(img)
    q = dev(e[:, :2].copy())
    out = dev(np.zeros((e.shape[0], 2), np.uint32))
    ptrs, cnts, keep = idx.stage_blocks([dev(b) for b in blocks])
    idx.cycle(q, out, ptrs, cnts, len(keep), insert_hint=e.shape[0])
    torch.cuda.synchronize()
    assert np.array_equal(idx.table_numpy(), exp)
    assert np.array_equal(host(out), O.search(ogeom(g), img, e[:, :2].copy()))


def test_insert_tail_fallback_beyond_lds_sort():
    """> 8192 leftovers after the parallel rounds: the in-order tail walks the
    request state array instead of sorting in LDS."""
    set_tuning("tail_rounds", 0)
    set_tuning("dup_groups", 0)                               # (round 2 would settle the duplicates as four groups)
    try:
        g = Geometry(mem_p=16)
        rng = np.random.default_rng(8)
        e = random_requests(g, rng, 20000, 1 << 30, hot=1.0)      # 4 keys, 20000 requests
        got, exp, info = run_insert_case(g, np.zeros((g.nbuckets, 16), np.uint32), [e])
        assert np.array_equal(got, exp), (first_diff(got, exp), info)
        assert info["tail_serial"] > 8192
    finally:
        set_tuning("dup_groups", 1)
    # the same batch with the duplicates settled as groups in round 2: nothing left for the tail
    got, exp, info = run_insert_case(g, np.zeros((g.nbuckets, 16), np.uint32), [e])
    assert np.array_equal(got, exp), (first_diff(got, exp), info)
    assert info["tail_serial"] == 0 and info["tail_parallel"] < 64


def test_insert_many_batches_grow_load_factor():
    """Repeated batches into one table up to LF 0.9 (cuckoo): every intermediate
    table image equals the oracle's."""
    g = Geometry(mem_p=16)
    og = ogeom(g)
    rng = np.random.default_rng(9)
    idx = HashIndex(g)
    exp = np.zeros((g.nbuckets, 16), np.uint32)
    for step in range(9):
        e = random_requests(g, rng, g.nbuckets * 8 // 10, 1 << 31)
        blocks = split_blocks(e, g, rng, "region")
        O.insert(og, exp, blocks)
        idx.insert([dev(b) for b in blocks])
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (step, first_diff(got, exp), idx.last_update_info())


def test_insert_test_properties_on_gpu():
    """libgpuhash/test/insert_test.c P1 (:178-195) and P2 (:237-244) with the
    reference test's own parameters: 16384 requests per pass, region-confined
    blocks, zeroed table."""
    g = Geometry(mem_p=24)
    idx = HashIndex(g)
    rng = np.random.default_rng(10)
    region = g.nbuckets // 8
    for _ in range(6):
        blocks = []
        for b in range(8):
            e = np.empty((2048, 3), np.uint32)
            e[:, 0] = rng.integers(1, 2**31, size=2048)
            e[:, 1] = b * region + rng.integers(0, region, size=2048)
            e[:, 2] = rng.integers(1, 2**31, size=2048)
            blocks.append(e)
        allreq = np.concatenate(blocks)
        idx.insert([dev(b) for b in blocks])
        out = host(idx.search(dev(allreq[:, :2].copy())))
        assert ((out[:, 0] == allreq[:, 2]) | (out[:, 1] == allreq[:, 2])).all()
        idx.delete(dev(allreq))
        out = host(idx.search(dev(allreq[:, :2].copy())))
        assert not ((out[:, 0] == allreq[:, 2]) | (out[:, 1] == allreq[:, 2])).any()
        idx.insert([dev(b) for b in blocks])


def test_survey_recorded_eviction_on_gpu():
    """SURVEY 8(c)(1) at the real MEM_P=30 geometry: victim 0x1003 moves from
    bucket 0x206 to 0x1006 = inserter_hash ^ victim_sig."""
    g = Geometry(mem_p=30)
    idx = HashIndex(g)
    t = idx.table.view(-1, 16)
    for b, s0, l0 in ((5, 0x5000, 100), (0x206, 0x1000, 200)):
        t[b, :8] = torch.arange(s0, s0 + 8, dtype=torch.int32, device="cuda")
        t[b, 8:] = torch.arange(l0, l0 + 8, dtype=torch.int32, device="cuda")
    idx.insert([dev(np.array([[0x203, 5, 777]], np.uint32))])
    torch.cuda.synchronize()
    assert t[0x206, 3].item() == 0x203 and t[0x206, 11].item() == 777
    assert t[0x1006, 3].item() == 0x1003 and t[0x1006, 11].item() == 203
    assert int(t[0x1205].abs().sum().item()) == 0
    assert idx.last_update_info()["first_hard"] == 0


# ----------------------------------------------------------------- cycle ----
@pytest.mark.parametrize("memset,overlap", [(False, "fused"), (True, "fused"), (False, "side"), (True, "side"),
                                            (False, "none"), (True, "none")])
@pytest.mark.parametrize("scheme", SCHEMES)
def test_cycle_equals_search_delete_insert(memset, overlap, scheme):
    """mkv_cycle (one call: clear -> search -> delete -> insert, mark pass beside the
    search kernel) against the oracle running the three steps in order, over
    several consecutive cycles on one stream."""
    g = Geometry(mem_p=16, scheme=scheme)
    og = ogeom(g)
    rng = np.random.default_rng(int(memset) * 2 + len(overlap) + 10 * scheme)
    img = random_table(g, rng, 0.5, 1 << 20)
    idx = HashIndex(g)
    idx.load_table(img)
    exp = img.copy()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for step in range(4):
            q = random_requests(g, rng, 5000, 1 << 20)[:, :2].copy()
            d = random_requests(g, rng, 700, 1 << 20)
            e = random_requests(g, rng, 4000, 1 << 20, hot=0.1)
            # make some searches/deletes hit: reuse stored signatures
            occ = np.argwhere(exp[:, :8] != 0)[:: max(1, len(np.argwhere(exp[:, :8] != 0)) // 2000)]
            for k, (b, s) in enumerate(occ[:2000]):
                q[k] = (exp[b, s], b)
                if k < 300:
                    d[k] = (exp[b, s], b, exp[b, 8 + s])
            blocks = split_blocks(e, g, rng, "region")
            dq, dd = dev(q), dev(d)
            dblocks = [dev(b) for b in blocks]
            ptrs, cnts, keep = idx.stage_blocks(dblocks)
            out = dev(np.full((q.shape[0], 2), 0x5555AAAA, np.uint32))
            idx.cycle(dq, out, ptrs, cnts, len(keep), delem=dd, memset=memset, overlap=overlap, stream=stream)
            exp_out = O.search(og, exp, q)
            O.delete(og, exp, d)
            O.insert(og, exp, blocks)
            stream.synchronize()
            assert np.array_equal(host(out), exp_out), f"search of cycle {step}"
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (step, first_diff(got, exp))


@pytest.mark.parametrize("scheme", [0, 1])
def test_cycle_on_pinned_host_buffers(scheme):
    """Requests, insert blocks and results in pinned HOST memory (mapped into the device's
    address space): the cycle's kernels read and write them over PCIe, no copies.  Same
    results as the oracle, several cycles in a row."""
    g = Geometry(mem_p=16, scheme=scheme)
    og = ogeom(g)
    rng = np.random.default_rng(4100 + scheme)
    img = random_table(g, rng, 0.5, 1 << 20)
    idx = HashIndex(g)
    idx.load_table(img)
    exp = img.copy()
    stream = torch.cuda.Stream()

    def pin(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).pin_memory()

    with torch.cuda.stream(stream):
        for step in range(3):
            q = random_requests(g, rng, 6000, 1 << 20)[:, :2].copy()
            d = random_requests(g, rng, 500, 1 << 20)
            e = random_requests(g, rng, 4000, 1 << 20, hot=0.1)
            occ = np.argwhere(exp[:, :8] != 0)[:2000]
            for k, (b, s) in enumerate(occ):
                q[k] = (exp[b, s], b)
                if k < 200:
                    d[k] = (exp[b, s], b, exp[b, 8 + s])
            blocks = split_blocks(e, g, rng, "region")
            hq, hd = pin(q), pin(d)
            hblocks = [pin(b.reshape(-1, 3)) for b in blocks]
            ptrs, cnts, keep = idx.stage_blocks(hblocks)          # device array of HOST block pointers
            out = pin(np.full((q.shape[0], 2), 0x5555AAAA, np.uint32))
            idx.cycle(hq, out, ptrs, cnts, len(keep), delem=hd, stream=stream, insert_hint=e.shape[0])
            exp_out = O.search(og, exp, q)
            O.delete(og, exp, d)
            O.insert(og, exp, blocks)
            stream.synchronize()
            assert np.array_equal(out.numpy().view(np.uint32), exp_out), f"search of cycle {step}"
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (step, first_diff(got, exp))


@pytest.mark.parametrize("scheme", [0, 1])
def test_largest_geometry_mem_p_33(scheme):
    """MEM_P=33, the largest table the library takes (8 GiB, 2^27 buckets; word offsets reach
    2^31): insert with duplicates, search, delete, search -- against the oracle on a lazily
    committed host image.  The tables are compared on every bucket a request can have touched
    (both candidates of every request; eviction chains do not occur at this fill)."""
    if torch.cuda.mem_get_info()[0] < (10 << 30):
        pytest.skip("needs 10 GiB of free device memory")
    g = Geometry(mem_p=33, scheme=scheme)
    og = ogeom(g)
    rng = np.random.default_rng(3300 + scheme)
    exp = O.new_table(og)
    idx = HashIndex(g)
    n = 300000
    e = random_requests(g, rng, n, 1 << 31, hot=0.05)
    e[: n // 8, 1] |= np.uint32(0xFFFF0000)                    # a good share in the topmost buckets
    blocks = split_blocks(e, g, rng, "region")
    st = O.insert(og, exp, blocks)
    idx.insert([dev(b.reshape(-1, 3)) for b in blocks])
    P = e[:, 1] & np.uint32(g.hash_mask)
    A = (((e[:, 1] ^ e[:, 0]) & np.uint32(g.block_hash_mask)) | (e[:, 1] & ~np.uint32(g.block_hash_mask))) & np.uint32(g.hash_mask)
    rows = np.unique(np.concatenate([P, A, [0, g.nbuckets - 1]])).astype(np.int64)
    assert rows.max() == g.nbuckets - 1 and (rows >> 24).max() == 7

    def touched():
        return idx.table.view(-1, 16)[torch.from_numpy(rows).cuda()].cpu().numpy().view(np.uint32)

    got = touched()
    assert np.array_equal(got, exp[rows]), first_diff(got, exp[rows])
    q = np.ascontiguousarray(e[:, :2])
    assert np.array_equal(host(idx.search(dev(q))), O.search(og, exp, q))
    d = np.ascontiguousarray(e[::2])
    O.delete(og, exp, d)
    idx.delete(dev(d))
    got = touched()
    assert np.array_equal(got, exp[rows]), first_diff(got, exp[rows])
    assert np.array_equal(host(idx.search(dev(q))), O.search(og, exp, q))
    # nothing outside the touched buckets was written: the table holds exactly the oracle's entries
    assert int((idx.table.view(-1, 16)[:, :8] != 0).sum().item()) == int((exp[rows][:, :8] != 0).sum())
    del idx
    torch.cuda.empty_cache()


def test_baseline_config1_shape_through_the_library():
    """BASELINE configs[0] (the reference's LOCAL_TEST shape: MEM_P=25, two-choice, reference keying,
    pre-load of P-1 = 838 859 SETs in batches of <= 8 x 4096 binned the receiver's way, then uniform
    GET batches of 32 767 = batch_max_search_job - 1) through the library: table image equal to the
    oracle's, every GET hits in its primary bucket with loc == key (tests/test_workload.py runs the
    same on the oracle alone)."""
    import workload as W
    mem_p = 25
    g = Geometry(mem_p=mem_p, scheme=1)
    og = ogeom(g)
    exp = O.new_table(og)
    idx = HashIndex(g)
    P = W.preload_count(mem_p)
    keys = np.arange(1, P, dtype=np.uint32)
    for lo in range(0, keys.shape[0], 8 * 4095):
        e = W.make_ielem(keys[lo:lo + 8 * 4095], None, W.KEYING_REF)
        blocks = W.bin_inserts(e, mem_p, 3, W.BIN_APP)
        O.insert(og, exp, blocks)
        idx.insert([dev(b.reshape(-1, 3)) for b in blocks], hint=0)          # the plain ABI: no size hint
    got = idx.table_numpy()
    assert np.array_equal(got, exp), first_diff(got, exp)
    z = W.ZipfKeys(P - 2, 0.0, 21)
    blk_mask = (1 << (mem_p - 9)) - 1
    for _ in range(3):
        k = z.keys(32767)
        out = host(idx.search(dev(W.make_selem(k, W.KEYING_REF))))
        assert np.array_equal(out[:, 0], k)                                   # 100 % hits, loc = key
        assert np.array_equal(out[:, 1], np.where((k & blk_mask) == 0, k, 0))


def test_index_fixture_through_the_library():
    """tests/golden/index_kat.json replayed through the C ABI: the behaviours SURVEY 8(c) recorded
    from the reference's code and the committed regression vectors, compared with the committed
    rows / table digests / search results (no oracle involved)."""
    from tests.golden_util import load_index_kat, initial_rows, blocks_of, check_step
    for sc in load_index_kat():
        g = Geometry(mem_p=sc["mem_p"], scheme=sc["scheme"], iblock_p=sc["iblock_p"], max_cuckoo=sc["max_cuckoo"])
        idx = HashIndex(g)
        bi, rows = initial_rows(sc)
        tv = idx.table.view(-1, 16)
        if len(bi):
            tv[torch.from_numpy(bi).cuda()] = dev(rows)
        for k, (op, exp) in enumerate(zip(sc["ops"], sc["expected"])):
            out = None
            if op["op"] == "insert":
                idx.insert([dev(b) for b in blocks_of(op)])
            elif op["op"] == "delete":
                idx.delete(dev(np.array(op["requests"], dtype=np.uint32).reshape(-1, 3)))
            else:
                out = host(idx.search(dev(np.array(op["requests"], dtype=np.uint32).reshape(-1, 2))))
            torch.cuda.synchronize()
            check_step(sc["name"], k, exp, lambda i: host(tv[torch.from_numpy(i).cuda()]),
                       lambda: idx.table_numpy(), out)
        del idx


def test_large_preload_with_chunks_and_carries_matches_oracle():
    """24 M keys into a MEM_P=29 table (load factor 0.36) in 1 M-request calls: every
    call runs as 64 K chunks with a size hint, barriers appear as the table fills and
    their unresolved requests are carried from chunk to chunk.  Final table image and a
    search over all keys equal the oracle's."""
    import workload as W
    g = Geometry(mem_p=29)
    og = ogeom(g)
    idx = HashIndex(g)
    exp = O.new_table(og)
    n_keys = 24 * (1 << 20)
    chains = undone = 0
    for lo in range(1, n_keys, 1 << 20):
        k = np.arange(lo, lo + (1 << 20), dtype=np.uint32)
        blocks = W.bin_inserts(W.make_ielem(k, None, W.KEYING_MIX), g.mem_p, g.iblock_p, W.BIN_REGION)
        idx.insert([dev(b) for b in blocks])
        O.insert(og, exp, blocks)
        info = idx.last_update_info()
        chains += info["chains"]
        undone += info["undone"]
        assert info["total"] == 1 << 20
    got = idx.table_numpy()
    assert np.array_equal(got, exp), first_diff(got, exp)
    assert chains > 0                             # the interesting paths were really taken
    q = W.make_selem(np.arange(1, 1 << 20, dtype=np.uint32), W.KEYING_MIX)
    assert np.array_equal(host(idx.search(dev(q))), O.search(og, exp, q))


def test_cycle_is_hipgraph_replayable():
    """One mkv_cycle captured into a hipGraph and replayed on new requests placed in the
    same buffers: round stamps and request counts live on the device, nothing is
    allocated or synchronised inside the call, so every replay is a fresh, exact cycle."""
    g = Geometry(mem_p=16)
    og = ogeom(g)
    rng = np.random.default_rng(77)
    img = random_table(g, rng, 0.4, 1 << 24)
    idx = HashIndex(g)
    idx.load_table(img)
    exp = img.copy()
    n_q, n_s, cap = 6000, 3000, 3000
    stream = torch.cuda.Stream()
    idx.reserve(65536, stream)
    q_buf = torch.zeros((n_q, 2), dtype=torch.int32, device="cuda")
    out_buf = torch.zeros((n_q, 2), dtype=torch.int32, device="cuda")
    set_buf = torch.zeros((8, cap, 3), dtype=torch.int32, device="cuda")      # 8 fixed block slices
    ptrs = torch.tensor([set_buf[b].data_ptr() for b in range(8)], dtype=torch.int64).cuda()
    cnts = torch.zeros(8, dtype=torch.int32, device="cuda")

    def load(step_rng):
        q = random_requests(g, step_rng, n_q, 1 << 24)[:, :2].copy()
        e = random_requests(g, step_rng, n_s, 1 << 24, hot=0.1)
        blocks = split_blocks(e, g, step_rng, "region")
        q_buf.copy_(dev(q))
        c = []
        for b, blk in enumerate(blocks):
            assert blk.shape[0] <= cap
            if blk.shape[0]:
                set_buf[b, :blk.shape[0]].copy_(dev(blk))
            c.append(blk.shape[0])
        cnts.copy_(torch.tensor(c, dtype=torch.int32))
        return q, blocks

    # warm-up call outside capture (workspaces exist), then capture one cycle
    q, blocks = load(np.random.default_rng(1))
    with torch.cuda.stream(stream):
        idx.cycle(q_buf, out_buf, ptrs, cnts, 8, stream=stream, insert_hint=n_s)
    stream.synchronize()
    ref_out = O.search(og, exp, q)
    O.insert(og, exp, blocks)
    assert np.array_equal(host(out_buf), ref_out) and np.array_equal(idx.table_numpy(), exp)

    graph = torch.cuda.CUDAGraph()
    q, blocks = load(np.random.default_rng(2))
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        idx.cycle(q_buf, out_buf, ptrs, cnts, 8, stream=torch.cuda.current_stream(), insert_hint=n_s)
    for rep in range(4):
        if rep:
            q, blocks = load(np.random.default_rng(10 + rep))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        ref_out = O.search(og, exp, q)
        O.insert(og, exp, blocks)
        assert np.array_equal(host(out_buf), ref_out), f"replay {rep}: search"
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (rep, first_diff(got, exp))


def test_round_stamps_running_low_clear_the_reservation_tables():
    """The reservation tables carry 40-bit round stamps that only grow.  A workspace started
    close to the end of that range (tests-only tuning) must ask the host to clear its tables
    (flag written by the tail kernel, acted on by a later call) and carry on with identical
    results -- here across a few dozen insert/delete calls, most of them with chains.
    (Diagnostics build: the product library cannot be told where to start its stamps.)"""
    L = load_library(diagnostics=True)
    g = Geometry(mem_p=14, scheme=0)
    og = ogeom(g)
    L.mkv_release_all()
    L.mkv_set_tuning(b"debug_first_stamp", (1 << 40) - (1 << 24) - 40)
    try:
        rng = np.random.default_rng(99)
        exp = random_table(g, rng, 0.8, 1 << 20)
        idx = HashIndex(g, lib=L)
        idx.load_table(exp)
        for step in range(40):
            e = random_requests(g, rng, int(rng.integers(200, 3000)), 1 << 20, hot=0.1)
            if step % 5 == 4:
                O.delete(og, exp, e)
                idx.delete(dev(e))
            else:
                blocks = split_blocks(e, g, rng, "region")
                O.insert(og, exp, blocks)
                idx.insert([dev(b) for b in blocks])
            got = idx.table_numpy()
            assert np.array_equal(got, exp), (step, first_diff(got, exp))
        from megakv_amd.index import _stream_ptr
        assert L.mkv_debug_stamp_resets(_stream_ptr(None), 0) >= 1      # the guard really fired
    finally:
        L.mkv_set_tuning(b"debug_first_stamp", 0)
        L.mkv_release_all()
