"""Big insert passes: partition into bins, apply per bucket with whole-bucket write-back, left-overs through
round 2 and the tail (megakv_amd/csrc/mkv_upd_apply.h).  The path is taken by passes announced with at least
Tuning::sorted_min requests (0 = never by default: the path is exact but not faster than the reservation
rounds); here the tuning forces it at every size, so that small
geometries -- where buckets are shared heavily, chains are long and the oracle finishes in no time -- exercise
it: the whole table against the oracle after every call, both schemes, through mkv_insert_hinted and through
mkv_cycle (search + delete + insert, the partition riding in the search launch).

Reference semantics: libgpuhash/gpu_hash.cu:277-323 (primary bucket: duplicate before empty, preferred slot),
:331-425 (alternate bucket and the eviction chain), :66-67 (closed regions)."""
import numpy as np
import pytest
import torch

from megakv_amd import Geometry, HashIndex, SCHEME_CUCKOO, SCHEME_2CHOICE
from megakv_amd.lib import load_library
from oracle import oracle as O
from tests.gpu_util import dev, host, ogeom, random_table, random_requests, split_blocks, first_diff

pytestmark = pytest.mark.gpu
SCHEMES = [SCHEME_CUCKOO, SCHEME_2CHOICE]


@pytest.fixture(autouse=True)
def _forced():
    L = load_library()
    assert L.mkv_set_tuning(b"sorted_min", 1) == 0
    assert L.mkv_set_tuning(b"sorted_cooldown", 0) == 0       # (never back off: every pass here takes the path)
    assert L.mkv_set_tuning(b"insert_chunk", 524288) == 0     # (one pass per call: other suites of a run leave it smaller)
    yield
    assert L.mkv_set_tuning(b"sorted_min", 0) == 0
    assert L.mkv_set_tuning(b"sorted_cooldown", 32) == 0


def sorted_stats():
    """{passes, fallbacks, left_over} of the device's insert workspace, None before there is one"""
    import ctypes
    out = (ctypes.c_uint32 * 4)()
    if load_library().mkv_sorted_stats(None, 0, ctypes.byref(out)) != 0:
        return None
    return {"passes": int(out[0]), "fallbacks": int(out[1]), "left_over": int(out[2])}


def insert_and_compare(g, img, blocks, what=""):
    exp = img.copy()
    O.insert(ogeom(g), exp, blocks)
    idx = HashIndex(g)
    idx.load_table(img)
    idx.insert([dev(b.reshape(-1, 3)) for b in blocks])
    got = idx.table_numpy()
    assert np.array_equal(got, exp), (what, first_diff(got, exp), idx.last_update_info())
    idx.stats = idx.sorted_stats()
    return idx


CASES = [
    # mem_p, n, fill, sig_space, hot, zero_low, zeros, blocks
    (12, 1, 0.0, 1000, 0.0, 0.0, 0.0, "single"),
    (12, 7, 0.5, 1000, 0.0, 0.0, 0.0, "region"),
    (12, 300, 0.0, 50, 0.0, 0.0, 0.0, "region"),        # 64 buckets, one bucket a bin: runs of many requests, duplicate signatures
    (12, 2000, 0.6, 100000, 0.1, 0.05, 0.02, "app"),     # left-overs and eviction chains everywhere, signature 0
    (14, 5000, 0.9, 100000, 0.0, 0.0, 0.0, "uneven"),    # nearly full: almost everything is left over
    (16, 20000, 0.3, 1 << 30, 0.3, 0.0, 0.0, "region"),  # hot duplicate keys: a bin overflows -> the pass falls back
    (16, 20000, 0.5, 1 << 30, 0.0, 0.2, 0.01, "app"),    # alternate == primary
    (18, 30000, 0.2, 1 << 31, 0.0, 0.0, 0.0, "region"),
    (20, 60000, 0.2, 1 << 31, 0.0, 0.0, 0.0, "region"),
    (20, 200000, 0.35, 1 << 31, 0.0, 0.0, 0.0, "app"),   # a few thousand left-overs, late alternates above applied requests
    (20, 60000, 0.85, 1 << 31, 0.05, 0.0, 0.0, "single"),
    (22, 200000, 0.8, 1 << 31, 0.0, 0.0, 0.0, "region"),
    (22, 400000, 0.5, 1 << 31, 0.02, 0.01, 0.0, "app"),
    (24, 524288, 0.2, 1 << 31, 0.0, 0.0, 0.0, "region"),  # the shape of BASELINE configs[3]'s insert half at 1/64 of the table
]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"p{c[0]}-n{c[1]}-f{c[2]}-{c[7]}")
def test_big_pass_matches_oracle(case, scheme):
    mem_p, n, fill, sig_space, hot, zero_low, zeros, how = case
    g = Geometry(mem_p=mem_p, scheme=scheme)
    rng = np.random.default_rng(hash((mem_p, n, scheme, 6)) & 0xFFFF)
    img = random_table(g, rng, fill, min(sig_space, 2**31))
    e = random_requests(g, rng, n, sig_space, hot, zero_low, zeros)
    before = sorted_stats()
    idx = insert_and_compare(g, img, split_blocks(e, g, rng, how), case)
    # the path under test was taken: the pass was binned (or, hot keys in one bin, fell back behind its partition)
    after = idx.stats
    # (a workspace that had to grow for this geometry starts its counts anew)
    ran = after["passes"] + after["fallbacks"]
    assert ran >= 1 and (before is None or ran in (1, before["passes"] + before["fallbacks"] + 1)), (before, after)
    if hot == 0.0:
        assert after["fallbacks"] == 0 or (before is not None and after["fallbacks"] == before["fallbacks"]), (before, after)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_consecutive_big_passes_fill_a_table(scheme):
    """Pass after pass of new keys from an empty table to load factor 0.75: the left-overs grow from none to most of
    a pass; every pass's table is the oracle's."""
    g = Geometry(mem_p=20, scheme=scheme)
    og = ogeom(g)
    rng = np.random.default_rng(61 + scheme)
    idx = HashIndex(g)
    exp = O.new_table(og)
    for step in range(12):
        e = random_requests(g, rng, 8192, 1 << 31)
        blocks = split_blocks(e, g, rng, "region")
        O.insert(og, exp, blocks)
        idx.insert([dev(b) for b in blocks])
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (step, first_diff(got, exp), idx.last_update_info())


def _bucket_requests(g, rng, bucket, n, sig_space=1 << 31):
    """n requests whose primary bucket is `bucket` (signatures distinct, non-zero)"""
    e = np.empty((n, 3), np.uint32)
    sigs = set()
    while len(sigs) < n:
        sigs.add(int(rng.integers(1 << 12, sig_space)))
    e[:, 0] = np.array(sorted(sigs), dtype=np.uint64)[rng.permutation(n)]
    e[:, 1] = bucket | (rng.integers(0, 1 << (32 - (g.mem_p - 6)), size=n).astype(np.uint32) << np.uint32(g.mem_p - 6))
    e[:, 2] = rng.integers(1, 2**32, size=n, dtype=np.uint64)
    return e


def _alt(g, sig, hash_):
    return int((((hash_ ^ sig) & g.block_hash_mask) | (hash_ & ~g.block_hash_mask)) & g.hash_mask)


def _with_alternate(g, rng, primary, alternate):
    """one request with the given primary and alternate bucket (same closed region)"""
    assert (primary & ~g.block_hash_mask) == (alternate & ~g.block_hash_mask)
    low = (primary ^ alternate) & g.block_hash_mask
    sig = (int(rng.integers(1, 1 << 10)) << (g.mem_p - 6)) | low    # region-local bits of the signature = P ^ A
    sig &= 0xffffffff
    if sig == 0:
        sig = low | (1 << 31)
    e = np.array([[sig, primary, int(rng.integers(1, 2**32))]], dtype=np.uint64).astype(np.uint32)
    assert _alt(g, int(e[0, 0]), int(e[0, 1])) == alternate
    return e


@pytest.mark.parametrize("scheme", SCHEMES)
def test_left_over_request_meets_applied_requests_below_and_above_it(scheme):
    """NEXT.md case 1: request L finds its primary bucket P full and goes to its alternate bucket A, where requests
    of LOWER and of HIGHER index (primary bucket A) have been applied by k_apply: the later ones are an order
    violation -- they must be taken back and run again behind L.  Slot choice in A depends on that order."""
    g = Geometry(mem_p=16, scheme=scheme)
    rng = np.random.default_rng(7)
    P, A = 5, 9
    img = np.zeros((g.nbuckets, 16), np.uint32)
    img[P, :8] = np.arange(101, 109, dtype=np.uint32)        # P full of other keys
    img[P, 8:] = 7
    img[A, :5] = np.arange(201, 206, dtype=np.uint32)        # A: three free slots
    img[A, 8:13] = 9
    below = _bucket_requests(g, rng, A, 1)
    left = _with_alternate(g, rng, P, A)
    above = _bucket_requests(g, rng, A, 3)                   # with L and `below` that is five for three free slots
    e = np.concatenate([below, left, above])
    insert_and_compare(g, img, [np.ascontiguousarray(e)], "below, left-over, above")
    e2 = np.concatenate([above[:1], left, below, above[1:]])
    insert_and_compare(g, img, [np.ascontiguousarray(e2)], "another order")


def test_chain_walks_into_a_bucket_with_applied_requests():
    """NEXT.md case 2: a hard request (P and A full) evicts from A; the victim's alternate bucket V got requests of
    HIGHER index applied by k_apply -- violation, take-back, replay behind the chain."""
    g = Geometry(mem_p=16, scheme=SCHEME_CUCKOO)
    rng = np.random.default_rng(8)
    P, A = 17, 33
    hard = _with_alternate(g, rng, P, A)
    v = int(hard[0, 0]) & 7                                  # the victim slot: ORIGINAL signature & 7 (ref :360)
    img = np.zeros((g.nbuckets, 16), np.uint32)
    img[P, :8] = np.arange(1001, 1009, dtype=np.uint32)
    img[P, 8:] = 3
    img[A, :8] = np.arange(2001, 2009, dtype=np.uint32)
    img[A, 8:] = 4
    victim_sig = int(img[A, v])
    V = _alt(g, victim_sig, int(hard[0, 1]))                 # the victim moves with the INSERTER's hash (ref :334-335)
    assert V not in (P, A)
    img[V, :6] = np.arange(3001, 3007, dtype=np.uint32)      # V: two free slots
    img[V, 8:14] = 5
    later = _bucket_requests(g, rng, V, 3)                   # three later requests of primary bucket V: one too many
    earlier = _bucket_requests(g, rng, V, 1)
    e = np.concatenate([earlier, hard, later])
    insert_and_compare(g, img, [np.ascontiguousarray(e)], "chain into an applied bucket")
    filler = random_requests(g, rng, 3000, 1 << 31)
    e = np.concatenate([filler[:1000], earlier, filler[1000:2000], hard, later, filler[2000:]])
    insert_and_compare(g, img, [np.ascontiguousarray(e)], "the same among 3000 others")


@pytest.mark.parametrize("scheme", SCHEMES)
def test_duplicates_of_a_key_straddle_a_left_over_request(scheme):
    """NEXT.md case 3: key K comes three times in a pass on bucket B; between its second and third coming a request
    of another bucket overflows into B.  k_apply applies K once (a later duplicate would rewrite a word of this
    pass: left over, with everything behind it on B); the table's location of K is the last one's."""
    g = Geometry(mem_p=16, scheme=scheme)
    rng = np.random.default_rng(9)
    B, P = 70, 90
    img = np.zeros((g.nbuckets, 16), np.uint32)
    img[P, :8] = np.arange(501, 509, dtype=np.uint32)
    img[P, 8:] = 6
    img[B, :4] = np.arange(601, 605, dtype=np.uint32)
    img[B, 8:12] = 2
    k = _bucket_requests(g, rng, B, 1)
    dup = lambda loc: np.array([[k[0, 0], k[0, 1], loc]], np.uint32)
    other = _bucket_requests(g, rng, B, 2)
    left = _with_alternate(g, rng, P, B)
    e = np.concatenate([dup(11), other[:1], dup(22), left, dup(33), other[1:]])
    insert_and_compare(g, img, [np.ascontiguousarray(e)], "duplicates around a left-over")
    # ... and an existing key rewritten twice in one pass, with a new key between
    ex = np.array([[601, B, 77]], np.uint32)
    e = np.concatenate([ex, other[:1], np.array([[601, B, 88]], np.uint32), left, np.array([[601, B, 99]], np.uint32)])
    insert_and_compare(g, img, [np.ascontiguousarray(e)], "an existing key three times")


def test_signature_zero_requests_stay_with_the_tail():
    """signature 0 with a location "matches" every empty slot (ref :277-287): k_apply leaves such a request, and what
    follows it on its bucket, to round 2 and the tail"""
    g = Geometry(mem_p=14, scheme=SCHEME_CUCKOO)
    rng = np.random.default_rng(10)
    img = random_table(g, rng, 0.4, 1 << 20)
    e = random_requests(g, rng, 4000, 1 << 20, zeros=0.2)
    insert_and_compare(g, img, split_blocks(e, g, rng, "region"))


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("with_delete", [False, True])
def test_cycle_with_a_big_insert_pass(scheme, with_delete):
    """mkv_cycle: the partition rides in the search launch; the searches see the table as the cycle found it, the
    deletes run before the apply kernel reads the table."""
    g = Geometry(mem_p=20, scheme=scheme)
    og = ogeom(g)
    rng = np.random.default_rng(11 + scheme)
    idx = HashIndex(g)
    exp = O.new_table(og)
    prev = None
    for step in range(6):
        e = random_requests(g, rng, 30000, 1 << 31)
        blocks = split_blocks(e, g, rng, "region")
        q = np.ascontiguousarray((prev if prev is not None else e)[:20000, :2])
        want = O.search(og, exp, q)
        d = None
        if with_delete and prev is not None:
            d = np.ascontiguousarray(prev[::5])
            O.delete(og, exp, d)
        O.insert(og, exp, blocks)
        dq = dev(q)
        out = torch.empty((q.shape[0], 2), dtype=torch.int32, device="cuda")
        ptrs, cnts, keep = idx.stage_blocks([dev(b) for b in blocks])
        dd = dev(d) if d is not None else None
        idx.cycle(dq, out, ptrs, cnts, len(keep), delem=dd, insert_hint=e.shape[0])
        torch.cuda.synchronize()
        assert np.array_equal(host(out), want), f"search of step {step}"
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (step, first_diff(got, exp), idx.last_update_info(cycle_insert=True))
        prev = e


def test_hot_key_pass_falls_back_and_the_next_pass_is_binned_again():
    """A pass of 60 000 duplicates of four keys overflows a bin (kSortBinCap): the device finds out, the pass runs
    the old way behind its partition (mark, rounds), and nothing of it lingers: the passes after it are exact."""
    g = Geometry(mem_p=22, scheme=SCHEME_CUCKOO)
    og = ogeom(g)
    rng = np.random.default_rng(12)
    idx = HashIndex(g)
    exp = O.new_table(og)
    for step, hot in enumerate([0.0, 0.9, 0.0, 0.0, 0.95, 0.0]):
        e = random_requests(g, rng, 66000, 1 << 31, hot=hot)
        blocks = split_blocks(e, g, rng, "region")
        O.insert(og, exp, blocks)
        idx.insert([dev(b) for b in blocks])
        got = idx.table_numpy()
        assert np.array_equal(got, exp), (step, hot, first_diff(got, exp), idx.last_update_info())
