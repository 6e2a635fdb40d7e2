"""Pins the CPU oracle against everything the reference's own tests hold for
the libgpuhash path: the two known-answer table images, the two
properties of insert_test.c, and the behaviours SURVEY recorded from the
reference source.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O


def rng(seed):
    return np.random.default_rng(seed)


# ---- masks (reference gpu_hash.h:57-69) -----------------------------------
@pytest.mark.parametrize("mem_p", [12, 20, 25, 30, 33])
def test_masks_match_header_formulas(mem_p):
    g = O.geom(mem_p=mem_p)
    hash_mask = (1 << (mem_p - 6)) - 1
    blk_mask = (1 << (mem_p - 6 - 3)) - 1
    r = rng(mem_p)
    for h, s in r.integers(0, 2**32, size=(200, 2), dtype=np.uint64):
        h, s = int(h), int(s)
        assert O.primary(g, h) == h & hash_mask
        alt = (((h ^ s) & blk_mask) | (h & ~blk_mask & 0xFFFFFFFF)) & hash_mask
        assert O.alternate(g, h, s) == alt
        # both candidates share the region bits (top IBLOCK_P bits of the index)
        assert (O.primary(g, h) >> (mem_p - 9)) == (alt >> (mem_p - 9))


def test_default_geometry_constants():
    g = O.geom(mem_p=30)
    assert O.nbuckets(g) == 1 << 24
    assert O.primary(g, 0xFFFFFFFF) == 0xFFFFFF           # HASH_MASK
    assert O.alternate(g, 0, 0xFFFFFFFF) == 0x1FFFFF      # BLOCK_HASH_MASK


# ---- KAT 1: libgpuhash/test/back/py_search_stream.c:105-129 ---------------
def kat_search_image(g):
    """u32 index i: even 8-groups (signature rows) <- i % 8 + 1, odd (location
    rows) <- 1   (py_search_stream.c:105-114)"""
    n = O.nbuckets(g)
    t = np.empty((n, 16), dtype=np.uint32)
    t[:, :8] = np.arange(1, 9, dtype=np.uint32)
    t[:, 8:] = 1
    return t


@pytest.mark.parametrize("mem_p", [16, 24])
def test_kat_search_image(mem_p):
    g = O.geom(mem_p=mem_p)
    t = kat_search_image(g)
    r = rng(1)
    n = 20000
    q = np.empty((n, 2), dtype=np.uint32)
    q[:, 0] = r.integers(1, 9, size=n)                       # sig in [1,8]   (:129)
    q[:, 1] = r.integers(0, O.nbuckets(g), size=n)           # hash in [0,BUC_NUM) (:128)
    out = O.search(g, t, q)
    assert (out == 1).all()                                   # hit in both buckets


# ---- KAT 2: libgpuhash/test/back/py_delete_stream.c:100-117 ---------------
def test_kat_delete_image():
    g = O.geom(mem_p=16)
    n = O.nbuckets(g)
    t = np.zeros((n, 16), dtype=np.uint32)
    t[:, :8] = np.arange(1, 9, dtype=np.uint32)              # sig[j] = j+1, loc = 0
    r = rng(2)
    # one delete per (bucket, sig) so the expected image is the closed form
    buckets = r.permutation(n)[:300]
    sigs = r.integers(1, 9, size=300)
    d = np.stack([sigs, buckets, np.zeros(300)], axis=1).astype(np.uint32)
    # keep only requests whose alternate bucket is not some other request's primary
    expect = t.copy()
    expect[buckets, sigs - 1] = 0
    O.delete(g, t, d)
    assert np.array_equal(t, expect)      # slot sig-1 of the primary zeroed, alternate untouched


def test_delete_repeat_goes_to_alternate():
    """Serial consequence of the early-continue (gpu_hash.cu:465-468): a second
    identical delete finds nothing in the primary and clears the alternate."""
    g = O.geom(mem_p=16)
    t = np.zeros((O.nbuckets(g), 16), dtype=np.uint32)
    t[:, :8] = np.arange(1, 9, dtype=np.uint32)
    d = np.array([[3, 100, 0], [3, 100, 0]], dtype=np.uint32)
    O.delete(g, t, d)
    alt = O.alternate(g, 100, 3)
    assert alt != 100
    assert t[100, 2] == 0 and t[alt, 2] == 0
    assert t[alt, 10] == 0 and (t[100, [0, 1, 3, 4, 5, 6, 7]] != 0).all()


# ---- properties of libgpuhash/test/insert_test.c --------------------------
@pytest.mark.parametrize("scheme", [O.SCHEME_CUCKOO, O.SCHEME_2CHOICE])
def test_insert_test_properties(scheme):
    """P1 (:178-195): every inserted (sig,hash) is found with its loc in out[2i]
    or out[2i+1].  P2 (:237-244): after deleting the same triples none is found.
    Input discipline as in the test: zeroed table, per-block hashes confined to
    the block's region (:117-124), load stops at 0.1 (:111)."""
    g = O.geom(mem_p=20, scheme=scheme)
    nb = O.nbuckets(g)
    t = O.new_table(g)
    r = rng(3)
    per_blk = 256
    region = nb // 8
    has = 0
    while has < 0.1 * nb * 8:
        blocks = []
        for b in range(8):
            e = np.empty((per_blk, 3), dtype=np.uint32)
            e[:, 0] = r.integers(1, 2**31, size=per_blk)                 # rand() signature
            e[:, 1] = b * region + r.integers(0, region, size=per_blk)   # lower_bond + rand() % ...
            e[:, 2] = r.integers(1, 2**31, size=per_blk)
            blocks.append(e)
        allreq = np.concatenate(blocks)
        O.insert(g, t, blocks)
        out = O.search(g, t, np.ascontiguousarray(allreq[:, :2]))
        found = (out[:, 0] == allreq[:, 2]) | (out[:, 1] == allreq[:, 2])
        assert found.all(), "P1 violated"
        O.delete(g, t, allreq)
        out = O.search(g, t, np.ascontiguousarray(allreq[:, :2]))
        again = (out[:, 0] == allreq[:, 2]) | (out[:, 1] == allreq[:, 2])
        assert not again.any(), "P2 violated"
        # put them back so the load factor really grows, like the reference loop
        O.insert(g, t, blocks)
        has += 8 * per_blk


# ---- behaviours recorded in SURVEY 8(c) (1)-(6) ---------------------------
def full_bucket(t, b, base_sig, base_loc):
    t[b, :8] = np.arange(base_sig, base_sig + 8, dtype=np.uint32)
    t[b, 8:] = np.arange(base_loc, base_loc + 8, dtype=np.uint32)


def test_cuckoo_victim_relocation_uses_inserter_hash():
    """(1): inserter hash 5 / sig 0x203 finds buckets 5 and 0x206 full; the victim
    0x1003 in slot 3 of 0x206 moves to 0x1006 = 5 ^ 0x1003, NOT to its own
    alternate 0x1205, and a later search for it misses (gpu_hash.cu:334-335)."""
    g = O.geom(mem_p=30)
    t = O.new_table(g)
    full_bucket(t, 5, 0x5000, 100)
    full_bucket(t, 0x206, 0x1000, 200)        # slot 3 holds sig 0x1003, loc 203
    st = O.insert(g, t, [np.array([[0x203, 5, 777]], dtype=np.uint32)])
    assert (t[0x206, 3], t[0x206, 11]) == (0x203, 777)
    assert (t[0x1006, 3], t[0x1006, 11]) == (0x1003, 203)   # preferred slot 3 = orig sig & 7
    assert t[0x1205].sum() == 0
    assert st["evictions"] == 1 and st["buckets_visited"] == 3
    # the victim lived in 0x206 as primary (hash 0x206): now unreachable
    out = O.search(g, t, np.array([[0x1003, 0x206]], dtype=np.uint32))
    assert (out == 0).all()
    out = O.search(g, t, np.array([[0x203, 5]], dtype=np.uint32))
    assert out[0, 0] == 0 and out[0, 1] == 777


def test_cuckoo_chain_drops_victim_after_five_relocations():
    g = O.geom(mem_p=30)
    t = O.new_table(g)
    # every bucket of region 0 the chain can reach is full: fill a small closed set
    # by making all victims share sig pattern -> use a table where all buckets full
    g2 = O.geom(mem_p=12)                    # 64 buckets, 8 regions of 8
    t2 = np.zeros((64, 16), dtype=np.uint32)
    for b in range(64):
        full_bucket(t2, b, 0x100 * (b + 1), 0x1000 * (b + 1))
    before = int((t2[:, :8] != 0).sum())
    st = O.insert(g2, t2, [np.array([[0xABC5, 3, 42]], dtype=np.uint32)])
    assert st["evictions"] == 5 and st["dropped"] == 1
    assert st["buckets_visited"] == 7          # primary + 6 alternates
    assert int((t2[:, :8] != 0).sum()) == before   # one in, one lost
    assert st["dword_stores"] == 12


def test_2choice_overflow_keeps_stale_loc():
    """(2)/(B-4): no room in either bucket: sig alone replaces slot sig&7 of the
    alternate, the old loc stays (gpu_hash.cu:197-209)."""
    g = O.geom(mem_p=30, scheme=O.SCHEME_2CHOICE)
    t = O.new_table(g)
    full_bucket(t, 5, 0x5000, 100)
    full_bucket(t, 0x206, 0x1000, 200)
    O.insert(g, t, [np.array([[0x203, 5, 777]], dtype=np.uint32)])
    assert (t[0x206, 3], t[0x206, 11]) == (0x203, 203)
    assert np.array_equal(t[5, :8], np.arange(0x5000, 0x5008))


def test_slot_preference_order():
    """(3): sig & 7 = 6 -> successive inserts take slots 6, 7, 0, 1."""
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    taken = []
    for i in range(4):
        sig = 0x100 * (i + 1) + 6
        O.insert(g, t, [np.array([[sig, 9, i + 1]], dtype=np.uint32)])
        taken.append(int(np.nonzero(t[9, :8] == sig)[0][0]))
    assert taken == [6, 7, 0, 1]


def test_duplicate_overwrites_loc_in_place():
    """(4)"""
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    O.insert(g, t, [np.array([[77, 9, 1], [77, 9, 2]], dtype=np.uint32)])
    assert int((t[9, :8] == 77).sum()) == 1
    assert t[9, 8 + int(np.nonzero(t[9, :8] == 77)[0][0])] == 2


def test_delete_needs_sig_and_loc_and_leaves_loc():
    """(5)"""
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    O.insert(g, t, [np.array([[77, 9, 5]], dtype=np.uint32)])
    O.delete(g, t, np.array([[77, 9, 6]], dtype=np.uint32))
    assert (t[9, :8] == 77).any()
    O.delete(g, t, np.array([[77, 9, 5]], dtype=np.uint32))
    assert not (t[9, :8] == 77).any()
    assert (t[9, 8:] == 5).any()            # stale loc stays


def test_search_sig_zero_matches_empty_slots():
    """(6): highest matching slot is the one that stays under serial lane order."""
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    O.insert(g, t, [np.array([[77, 9, 5]], dtype=np.uint32)])
    O.delete(g, t, np.array([[77, 9, 5]], dtype=np.uint32))
    slot = int(np.nonzero(t[9, 8:] == 5)[0][0])
    out = O.search(g, t, np.array([[0, 9]], dtype=np.uint32))
    assert out[0, 0] == (5 if slot == 7 else 0)
    t[9, 15] = 99
    out = O.search(g, t, np.array([[0, 9]], dtype=np.uint32))
    assert out[0, 0] == 99


def test_duplicate_check_precedes_empty_and_primary_reinsert():
    """(B-5): a key living only in its alternate bucket is re-inserted into the
    primary when the primary has room -> two copies."""
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    sig, h = 0x155, 9
    alt = O.alternate(g, h, sig)
    t[alt, 0], t[alt, 8] = sig, 1
    O.insert(g, t, [np.array([[sig, h, 2]], dtype=np.uint32)])
    assert (t[9, :8] == sig).sum() == 1 and t[alt, 8] == 1
    out = O.search(g, t, np.array([[sig, h]], dtype=np.uint32))
    assert tuple(out[0]) == (2, 1)


def test_all_zero_request_is_skipped():
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    st = O.insert(g, t, [np.array([[0, 9, 0], [0, 9, 4]], dtype=np.uint32)])
    assert st["skipped_zero"] == 1
    # sig 0 with a loc: the "duplicate" test matches an empty slot and writes loc only
    assert t[9, :8].sum() == 0 and t[9, 8] == 4


def test_search_untouched_on_miss():
    g = O.geom(mem_p=20)
    t = O.new_table(g)
    out = np.full((3, 2), 0xDEADBEEF, dtype=np.uint32)
    O.search(g, t, np.array([[1, 2], [3, 4], [5, 6]], dtype=np.uint32), out=out)
    assert (out == 0xDEADBEEF).all()


def test_search_mt_equals_single_thread():
    g = O.geom(mem_p=18)
    r = rng(9)
    t = r.integers(0, 50, size=(O.nbuckets(g), 16), dtype=np.uint32)
    q = r.integers(0, 2**32, size=(5000, 2), dtype=np.uint64).astype(np.uint32)
    q[:, 0] %= 50
    assert np.array_equal(O.search(g, t, q), O.search(g, t, q, threads=4))


def test_region_parallel_insert_equals_canonical_insert():
    """orc_insert_mt (CPU baseline only: one host thread per region-binned block) gives the
    table of the canonical single-thread insert; blocks that are not binned by region are
    refused (and run single-threaded)."""
    import numpy as np
    import workload as W
    g = O.geom(mem_p=18, scheme=O.SCHEME_CUCKOO)
    rng = np.random.default_rng(3)
    a, b = O.new_table(g), O.new_table(g)
    for _ in range(4):                                        # fill up until chains run
        keys = rng.integers(1, 1 << 30, size=9000, dtype=np.uint64).astype(np.uint32)
        blocks = W.bin_inserts(W.make_ielem(keys, None, W.KEYING_MIX), 18, 3, W.BIN_REGION)
        O.insert(g, a, blocks)
        assert O.insert_mt(g, b, blocks) is True
        assert np.array_equal(a, b)
    mixed = [np.concatenate(blocks[:2]), blocks[2]]            # first block spans two regions
    O.insert(g, a, mixed)
    assert O.insert_mt(g, b, mixed) is False
    assert np.array_equal(a, b)


# ---- the committed index fixture (tests/golden/index_kat.json) ---------------------------
def test_oracle_reproduces_index_fixture():
    """Every scenario of the fixture -- the SURVEY 8(c) answers and the regression vectors --
    replayed with the oracle: rows / table digests / search results as committed."""
    from tests.golden_util import load_index_kat, initial_rows, blocks_of, check_step
    for sc in load_index_kat():
        g = O.geom(mem_p=sc["mem_p"], scheme=sc["scheme"], iblock_p=sc["iblock_p"], max_cuckoo=sc["max_cuckoo"])
        t = O.new_table(g)
        idx, rows = initial_rows(sc)
        if len(idx):
            t[idx] = rows
        for k, (op, exp) in enumerate(zip(sc["ops"], sc["expected"])):
            out = None
            if op["op"] == "insert":
                O.insert(g, t, blocks_of(op))
            elif op["op"] == "delete":
                O.delete(g, t, np.array(op["requests"], dtype=np.uint32).reshape(-1, 3))
            else:
                out = O.search(g, t, np.array(op["requests"], dtype=np.uint32).reshape(-1, 2))
            check_step(sc["name"], k, exp, lambda i: t[i], lambda: t, out)


def test_sender_key_match_restatement_uses_strncmp_semantics():
    """orc_resolve_verify restates src/mega_send.c:408-414 (pick) and :543-553 (KEY_MATCH) and calls
    libc's strncmp as the reference does: keys that agree up to and including their first NUL are
    equal to it, whatever follows -- the reference's own 8-byte keys {k, (k&7)<<29|k}
    (src/mega_recv.c:700-704) are full of NULs."""
    arena = np.zeros((4, 40), np.uint8)
    arena[:, 10] = 8                                              # nkey (items.h: uint16 at byte 10)
    arena[1, 24:32] = [1, 0, 0, 0, 1, 0, 0, 0x20]                 # key id 1: {1, (1&7)<<29 | 1}
    arena[2, 24:32] = [1, 0, 5, 0, 9, 0, 0, 0]                    # differs only BEHIND the first NUL
    arena[3, 24:32] = [2, 0, 0, 0, 2, 0, 0, 0x40]
    arena[3, 10] = 9                                              # wrong nkey
    keys = np.array([[1, 0, 0, 0, 1, 0, 0, 0x20]] * 3 + [[2, 0, 0, 0, 2, 0, 0, 0x40]] * 2, np.uint8)
    pairs = np.array([[1, 0], [2, 0], [0, 1], [3, 0], [0, 0]], np.uint32)
    loc, hits, misses, fp = O.resolve_verify(pairs, keys, 8, arena, 40)
    assert loc.tolist() == [1, 2, 1, 0, 0] and (hits, misses, fp) == (3, 2, 1)
    loc, hits, misses, fp = O.resolve_verify(pairs, keys, 8, arena, 40, bytes_mode=True)
    assert loc.tolist() == [1, 0, 1, 0, 0] and (hits, misses, fp) == (2, 3, 2)
