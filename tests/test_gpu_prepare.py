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


@pytest.mark.parametrize("n,key_len", [(0, 8), (1, 8), (5000, 8), (200003, 16), (3001, 33)])
def test_resolve_with_key_match_like_the_sender(n, key_len):
    """src/mega_send.c:408-414 + 543-553 (KEY_MATCH): the location picked from the pair must
    name an item whose key length and key bytes equal the request's; else it is a miss.  The
    slab store is a flat arena here (reference item header: nkey at byte 10, key at byte 24,
    src/include/items.h:37-50).  Host restatement vs the device, incl. false positives (a
    location whose item holds another key), a wrong nkey, out-of-range locations and the
    reference's blind spot (a false positive in the primary bucket hides the alternate)."""
    L = load_library()
    rng = np.random.default_rng(n + key_len)
    items, stride, nkey_off, key_off = 4096, 64, 10, 24
    arena = rng.integers(0, 256, size=(items, stride), dtype=np.uint8)
    item_keys = rng.integers(0, 256, size=(items, key_len), dtype=np.uint8)
    arena[:, key_off:key_off + key_len] = item_keys
    arena[:, nkey_off] = key_len & 0xff
    arena[:, nkey_off + 1] = key_len >> 8
    wrong_len = rng.random(items) < 0.05
    arena[wrong_len, nkey_off] = (key_len + 1) & 0xff                  # same bytes, another length
    true_loc = rng.integers(1, items, size=n).astype(np.uint32)
    keys = np.zeros((n, key_len + 3), np.uint8)                        # (a stride wider than the key)
    keys[:, :key_len] = item_keys[true_loc] if n else 0
    pairs = np.zeros((n, 2), np.uint32)
    kind = rng.integers(0, 6, size=n)
    other = rng.integers(1, items, size=n).astype(np.uint32)
    pairs[kind == 0, 0] = true_loc[kind == 0]                           # hit in the primary bucket
    pairs[kind == 1, 1] = true_loc[kind == 1]                           # hit in the alternate bucket
    pairs[kind == 2, 0] = other[kind == 2]                              # false positive hiding ...
    pairs[kind == 2, 1] = true_loc[kind == 2]                           # ... the true entry (reference: a miss)
    pairs[kind == 3, 1] = other[kind == 3]                              # false positive (unless other == true)
    pairs[kind == 5, 0] = items + 77                                    # a location outside the arena
    # host restatement
    pick = np.where(pairs[:, 0] != 0, pairs[:, 0], pairs[:, 1])
    exp = np.zeros(n, np.uint32)
    fp = 0
    for i in range(n):
        l = int(pick[i])
        if l == 0:
            continue
        ok = l < items and (int(arena[l, nkey_off]) | int(arena[l, nkey_off + 1]) << 8) == key_len \
            and bytes(arena[l, key_off:key_off + key_len]) == bytes(keys[i, :key_len])
        if ok:
            exp[i] = l
        else:
            fp += 1
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_pairs = dev(pairs) if n else torch.zeros((1, 2), dtype=torch.int32, device="cuda")
    d_keys, d_arena = d(keys if n else np.zeros((1, key_len + 3), np.uint8)), d(arena)
    loc = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    ctr = torch.zeros(68, dtype=torch.int64, device="cuda")
    assert L.mkv_resolve_verify(_dev_ptr(d_pairs), n, _dev_ptr(d_keys), key_len + 3, key_len, _dev_ptr(d_arena), stride,
                                items, nkey_off, key_off, _dev_ptr(loc), _dev_ptr(ctr), _stream_ptr(None)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(host(loc)[:n], exp)
    c = ctr.cpu().numpy()
    assert (int(c[64]), int(c[65]), int(c[66])) == (int((exp != 0).sum()), n - int((exp != 0).sum()), fp)
    if n > 1000:
        assert fp > 0 and int(c[64]) > 0
