"""Device-side request preparation: key bytes -> (sig, hash)
as the reference's receiver derives them (src/mega_recv.c:349-362), stable
binning of insert requests into blocks (src/mega_recv.c:476-477 and
libgpuhash/test/insert_test.c:117-124), and the end-to-end path keys -> derive
-> bin -> gpu insert -> search, against host-side restatements and the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from megakv_amd import Geometry, HashIndex
from megakv_amd.index import _dev_ptr, _stream_ptr
from megakv_amd.lib import load_library
from oracle import oracle as O
from tests.gpu_util import dev, host, ogeom

pytestmark = pytest.mark.gpu


def derive_host(keys, key_len, fold):
    """restatement of src/mega_recv.c:349-362 on uint8[n, stride]"""
    n = keys.shape[0]
    sig = keys[:, :8].copy().view("<u8").reshape(n)
    if fold:
        off = 8
        while off <= key_len - 8:
            sig ^= keys[:, off:off + 8].copy().view("<u8").reshape(n)
            off += 8
        if off < key_len:
            tail = np.zeros((n, 8), np.uint8)
            tail[:, :key_len - off] = keys[:, off:key_len]
            sig ^= tail.view("<u8").reshape(n)
    return (sig & 0xFFFFFFFF).astype(np.uint32), (sig >> 32).astype(np.uint32)


@pytest.mark.parametrize("key_len,stride", [(8, 8), (16, 16), (32, 32), (13, 16), (128, 128), (20, 23)])
@pytest.mark.parametrize("fold", [0, 1])
def test_derive_matches_receiver_rule(key_len, stride, fold):
    L = load_library()
    rng = np.random.default_rng(key_len * 7 + fold)
    n = 10007
    keys = rng.integers(0, 256, size=(n, stride), dtype=np.uint8)
    locs = rng.integers(1, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    sig, hsh = derive_host(keys, key_len, fold)
    dk = torch.from_numpy(keys).cuda()
    dl = dev(locs)
    for words in (2, 3):
        out = torch.zeros((n, words), dtype=torch.int32, device="cuda")
        rc = L.mkv_derive_requests(_dev_ptr(dk), n, key_len, stride, fold, _dev_ptr(dl), _dev_ptr(out), words,
                                   _stream_ptr(None))
        assert rc == 0
        got = host(out)
        assert np.array_equal(got[:, 0], sig) and np.array_equal(got[:, 1], hsh)
        if words == 3:
            assert np.array_equal(got[:, 2], locs)


@pytest.mark.parametrize("words", [2, 3])
@pytest.mark.parametrize("rule", ["receiver", "region"])
@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 200003])
def test_route_requests_is_the_stable_binning(words, rule, n):
    L = load_library()
    g = Geometry(mem_p=30)
    rng = np.random.default_rng(n + words)
    req = rng.integers(0, 2**32, size=(n, words), dtype=np.uint64).astype(np.uint32)
    if rule == "receiver":
        mask, shift = 0xFFFFFFFF, 32 - g.iblock_p            # hash >> (32 - bits), src/mega_recv.c:476-477
    else:
        mask, shift = g.hash_mask, g.mem_p - 6 - g.iblock_p  # region of the bucket, insert_test.c:117-124
    blk = ((req[:, 1] & mask) >> shift).astype(np.int64)
    order = np.argsort(blk, kind="stable")
    d_in = dev(req)
    d_out = torch.zeros((n, words), dtype=torch.int32, device="cuda")
    perm = torch.zeros(n, dtype=torch.int32, device="cuda")
    counts = torch.zeros(8, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(((n + 1023) // 1024) * 8 + 64, dtype=torch.int32, device="cuda")
    rc = L.mkv_route_requests(_dev_ptr(d_in), words, n, mask, shift, 8, _dev_ptr(d_out), _dev_ptr(perm),
                              _dev_ptr(counts), _dev_ptr(scratch), scratch.numel() * 4, _stream_ptr(None))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(counts.cpu().numpy(), np.bincount(blk, minlength=8))
    assert np.array_equal(perm.cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(host(d_out), req[order])
    if words == 2:
        # undo: results routed back to request order
        res = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
        back = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
        assert L.mkv_unroute_results(_dev_ptr(dev(res)), _dev_ptr(perm), n, _dev_ptr(back), _stream_ptr(None)) == 0
        exp = np.empty_like(res)
        exp[order] = res
        assert np.array_equal(host(back), exp)


def test_keys_to_table_entirely_on_the_device():
    """16-byte keys -> derive -> bin the receiver's way -> gpu_hash insert -> search,
    all on the device; table and results equal the oracle fed by the host-side
    restatement of the same steps."""
    L = load_library()
    g = Geometry(mem_p=22)
    rng = np.random.default_rng(5)
    n = 60000
    keys = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
    locs = rng.integers(1, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    sig, hsh = derive_host(keys, 16, 1)
    req = np.stack([sig, hsh, locs], axis=1)
    blk = (hsh >> (32 - g.iblock_p)).astype(np.int64)
    blocks = [np.ascontiguousarray(req[blk == b]) for b in range(8)]
    exp = O.new_table(ogeom(g))
    O.insert(ogeom(g), exp, blocks)

    idx = HashIndex(g)
    dk, dl = torch.from_numpy(keys).cuda(), dev(locs)
    d_req = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    assert L.mkv_derive_requests(_dev_ptr(dk), n, 16, 16, 1, _dev_ptr(dl), _dev_ptr(d_req), 3, _stream_ptr(None)) == 0
    binned = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    perm = torch.zeros(n, dtype=torch.int32, device="cuda")
    counts = torch.zeros(8, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(((n + 1023) // 1024) * 8 + 64, dtype=torch.int32, device="cuda")
    assert L.mkv_route_requests(_dev_ptr(d_req), 3, n, 0xFFFFFFFF, 32 - g.iblock_p, 8, _dev_ptr(binned),
                                _dev_ptr(perm), _dev_ptr(counts), _dev_ptr(scratch), scratch.numel() * 4,
                                _stream_ptr(None)) == 0
    # block pointers / counts for the ABI: built on the device from the counts
    c = counts.to(torch.int64)
    offs = torch.cumsum(c, 0) - c
    ptrs = (binned.data_ptr() + offs * 12).to(torch.int64)
    cnts = c.to(torch.int32)
    idx.insert_staged(ptrs, cnts, 8, None, n)
    got = idx.table_numpy()
    assert np.array_equal(got, exp)
    d_q = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    assert L.mkv_derive_requests(_dev_ptr(dk), n, 16, 16, 1, ctypes.c_void_p(0), _dev_ptr(d_q), 2, _stream_ptr(None)) == 0
    out = idx.search(d_q)
    assert np.array_equal(host(out), O.search(ogeom(g), exp, np.ascontiguousarray(req[:, :2])))


@pytest.mark.parametrize("n", [0, 1, 1000, 300007])
def test_resolve_results_like_the_sender(n):
    """src/mega_send.c:408-414: loc = out[2i], else out[2i+1]; 0 = miss."""
    L = load_library()
    rng = np.random.default_rng(n)
    pairs = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    pairs[rng.random(n) < 0.4, 0] = 0
    pairs[rng.random(n) < 0.5, 1] = 0
    exp = np.where(pairs[:, 0] != 0, pairs[:, 0], pairs[:, 1]) if n else np.zeros(0, np.uint32)
    d_pairs = dev(pairs) if n else torch.zeros((1, 2), dtype=torch.int32, device="cuda")
    loc = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    ctr = torch.zeros(66, dtype=torch.int64, device="cuda")
    assert L.mkv_resolve_results(_dev_ptr(d_pairs), n, _dev_ptr(loc), _dev_ptr(ctr), _stream_ptr(None)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(host(loc)[:n], exp)
    c = ctr.cpu().numpy()
    assert int(c[64]) == int((exp != 0).sum()) and int(c[65]) == n - int((exp != 0).sum())


def _verify_on_device(pairs, keys, key_len, arena, stride, items, nkey_off, key_off, compare):
    L = load_library()
    n = pairs.shape[0]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_pairs = dev(pairs) if n else torch.zeros((1, 2), dtype=torch.int32, device="cuda")
    d_keys, d_arena = d(keys if n else np.zeros((1, keys.shape[1]), np.uint8)), d(arena)
    loc = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    ctr = torch.zeros(68, dtype=torch.int64, device="cuda")
    assert L.mkv_resolve_verify(_dev_ptr(d_pairs), n, _dev_ptr(d_keys), keys.shape[1], key_len, _dev_ptr(d_arena), stride,
                                items, nkey_off, key_off, compare, _dev_ptr(loc), _dev_ptr(ctr), _stream_ptr(None)) == 0
    torch.cuda.synchronize()
    c = ctr.cpu().numpy()
    return host(loc)[:n], (int(c[64]), int(c[65]), int(c[66]))


@pytest.mark.parametrize("compare", [0, 1], ids=["strncmp", "bytes"])
@pytest.mark.parametrize("n,key_len,pad", [(0, 8, 3), (1, 8, 0), (5000, 8, 0), (200003, 16, 3), (3001, 33, 3), (777, 130, 1)])
def test_resolve_with_key_match_like_the_sender(n, key_len, pad, compare):
    """src/mega_send.c:408-414 + 543-553 (KEY_MATCH): the location picked from the pair must
    name an item whose key length equals the request's and whose key compares equal -- with
    strncmp in the reference (compare 0: it stops at the first NUL), or byte for byte (compare
    1); else it is a miss.  The slab store is a flat arena here (reference item header: nkey at
    byte 10, key at byte 24, src/include/items.h:37-50).  Checked against the C restatement of
    those lines (oracle/gpuhash_oracle.c: orc_resolve_verify, which calls libc's strncmp), incl.
    false positives (a location whose item holds another key), a wrong nkey, out-of-range
    locations, the reference's blind spot (a false positive in the primary bucket hides the
    alternate), keys full of NUL bytes, and unaligned key strides."""
    rng = np.random.default_rng(n + key_len)
    items, stride, nkey_off, key_off = 4096, 24 + ((key_len + 15) // 16) * 16, 10, 24
    arena = rng.integers(0, 256, size=(items, stride), dtype=np.uint8)
    item_keys = rng.integers(0, 256, size=(items, key_len), dtype=np.uint8)
    item_keys[rng.random(item_keys.shape) < 0.15] = 0                   # NUL bytes anywhere
    arena[:, key_off:key_off + key_len] = item_keys
    arena[:, nkey_off] = key_len & 0xff
    arena[:, nkey_off + 1] = key_len >> 8
    wrong_len = rng.random(items) < 0.05
    arena[wrong_len, nkey_off] = (key_len + 1) & 0xff                  # same bytes, another length
    true_loc = rng.integers(1, items, size=n).astype(np.uint32)
    keys = np.zeros((n, key_len + pad), np.uint8)                      # (a stride wider than the key, unaligned)
    keys[:, :key_len] = item_keys[true_loc] if n else 0
    # some requests differ from their item only in ONE byte, before or after a NUL
    flip = rng.random(n) < 0.3
    pos = rng.integers(0, key_len, size=n)
    keys[flip, pos[flip]] ^= 0x5a
    pairs = np.zeros((n, 2), np.uint32)
    kind = rng.integers(0, 6, size=n)
    other = rng.integers(1, items, size=n).astype(np.uint32)
    pairs[kind == 0, 0] = true_loc[kind == 0]                           # hit in the primary bucket
    pairs[kind == 1, 1] = true_loc[kind == 1]                           # hit in the alternate bucket
    pairs[kind == 2, 0] = other[kind == 2]                              # false positive hiding ...
    pairs[kind == 2, 1] = true_loc[kind == 2]                           # ... the true entry (reference: a miss)
    pairs[kind == 3, 1] = other[kind == 3]                              # false positive (unless other == true)
    pairs[kind == 5, 0] = items + 77                                    # a location outside the arena
    exp, hits, misses, fp = O.resolve_verify(pairs, keys, key_len, arena, stride, nkey_off, key_off, bytes_mode=bool(compare))
    got, totals = _verify_on_device(pairs, keys, key_len, arena, stride, items, nkey_off, key_off, compare)
    assert np.array_equal(got, exp)
    assert totals == (hits, misses, fp)
    if n > 1000:
        assert fp > 0 and hits > 0


def test_key_match_on_the_reference_keys_differs_from_a_byte_compare():
    """The reference's LOCAL_TEST / benchmark keys are 8 bytes {k, (k & 7) << 29 | k}
    (src/mega_recv.c:700-704, benchmark/benchmark.c:495-499): little-endian k < 2^24 has NUL
    bytes from byte 1..3 on, so to strncmp (src/mega_send.c:544) every item whose key shares the
    bytes before the first NUL IS the key.  The device follows the reference on exactly that
    (compare 0) and tells the keys apart byte for byte on request (compare 1)."""
    items, stride, nkey_off, key_off, key_len = 1 << 16, 40, 10, 24, 8
    k = np.arange(items, dtype=np.uint32)
    item_keys = np.stack([k, ((k & 7) << 29) | k], axis=1).astype(np.uint32).view(np.uint8).reshape(items, 8)
    arena = np.zeros((items, stride), np.uint8)
    arena[:, key_off:key_off + 8] = item_keys
    arena[:, nkey_off] = key_len
    n = 50000
    rng = np.random.default_rng(3)
    want = rng.integers(1, items, size=n).astype(np.uint32)                 # the key the request asks for
    # the location a 32-bit signature collision would return: another key id with the same low byte
    # (and, for ids < 256, the same bytes up to the first NUL)
    coll = ((rng.integers(0, items >> 8, size=n).astype(np.uint32) << 8) | (want & 0xff)).astype(np.uint32)
    coll[coll == 0] = 256
    keys = item_keys[want]
    pairs = np.zeros((n, 2), np.uint32)
    pairs[:, 0] = np.where(rng.random(n) < 0.5, want, coll)
    for compare in (0, 1):
        exp, hits, misses, fp = O.resolve_verify(pairs, keys, key_len, arena, stride, nkey_off, key_off, bytes_mode=bool(compare))
        got, totals = _verify_on_device(pairs, keys, key_len, arena, stride, items, nkey_off, key_off, compare)
        assert np.array_equal(got, exp) and totals == (hits, misses, fp), compare
        if compare == 0:
            accepted_wrong = int(((got != 0) & (pairs[:, 0] != want)).sum())
        else:
            assert int(((got != 0) & (pairs[:, 0] != want)).sum()) == 0
    assert accepted_wrong > 0, "strncmp accepts keys that differ behind a NUL: the reference's behaviour"

