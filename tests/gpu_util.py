"""Helpers shared by the GPU parity tests: request/table generators and the
device <-> oracle comparison.  Everything goes through megakv_amd.HashIndex,
i.e. the C ABI of the HIP library."""
import numpy as np
import torch

from megakv_amd import Geometry, HashIndex
from oracle import oracle as O


def dev(a):
    """numpy uint32 array -> int32 CUDA tensor with the same bits"""
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint32)


def ogeom(g: Geometry):
    return O.geom(mem_p=g.mem_p, scheme=g.scheme, iblock_p=g.iblock_p, max_cuckoo=g.max_cuckoo)


def random_table(g: Geometry, rng, fill, sig_space):
    """Table image with ~fill of the slots occupied, signatures from [1, sig_space]."""
    nb = g.nbuckets
    t = np.zeros((nb, 16), dtype=np.uint32)
    occ = rng.random((nb, 8)) < fill
    t[:, :8] = np.where(occ, rng.integers(1, sig_space + 1, size=(nb, 8)), 0)
    t[:, 8:] = np.where(occ, rng.integers(1, 2**32, size=(nb, 8), dtype=np.uint64), 0)
    return t


def random_requests(g: Geometry, rng, n, sig_space, hot=0.0, zero_low=0.0, zeros=0.0, shape="random"):
    """[n,3] (sig, hash, loc).  hot: fraction drawn from 4 hot keys; zero_low:
    fraction whose signature has no region-local bits (alternate == primary);
    zeros: fraction of sig == 0 requests (half of them with loc == 0 too)."""
    e = np.empty((n, 3), dtype=np.uint32)
    e[:, 0] = rng.integers(1, sig_space + 1, size=n)
    e[:, 1] = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    e[:, 2] = rng.integers(1, 2**32, size=n, dtype=np.uint64)
    if shape == "ref" and n:
        # the reference's LOCAL_TEST keying on sequential ids: sig = k, hash = (k & 7) << 29 | k
        # (every key of a region has the same alternate bucket, src/mega_recv.c:700-704)
        k = (rng.integers(1, max(2, sig_space)) + np.arange(n, dtype=np.uint64)).astype(np.uint32)
        rng.shuffle(k)
        e[:, 0] = k
        e[:, 1] = ((k & np.uint32(7)) << np.uint32(29)) | k
    elif shape == "cluster" and n:
        # many different signatures on few primary buckets: long runs of registrants per bucket
        centres = rng.integers(0, 2**32, size=max(1, n // 40), dtype=np.uint64).astype(np.uint32)
        e[:, 1] = centres[rng.integers(0, len(centres), size=n)]
    if hot > 0 and n:
        hk = np.stack([rng.integers(1, sig_space + 1, size=4),
                       rng.integers(0, 2**32, size=4, dtype=np.uint64)], axis=1).astype(np.uint32)
        m = rng.random(n) < hot
        pick = rng.integers(0, 4, size=n)
        e[m, 0] = hk[pick[m], 0]
        e[m, 1] = hk[pick[m], 1]
    if zero_low > 0 and n:
        m = rng.random(n) < zero_low
        e[m, 0] &= ~np.uint32(g.block_hash_mask)
        e[m & (e[:, 0] == 0), 0] = np.uint32(g.block_hash_mask + 1)
    if zeros > 0 and n:
        m = rng.random(n) < zeros
        e[m, 0] = 0
        m2 = m & (rng.random(n) < 0.5)
        e[m2, 2] = 0
    return e


def split_blocks(e, g: Geometry, rng, how):
    n = e.shape[0]
    if how == "single":
        return [e]
    if how == "region":
        r = ((e[:, 1] & g.hash_mask) >> (g.mem_p - 6 - g.iblock_p)).astype(np.int64)
        return [np.ascontiguousarray(e[r == b]) for b in range(g.insert_blocks)]
    if how == "app":
        r = (e[:, 1] >> (32 - g.iblock_p)).astype(np.int64) if g.iblock_p else np.zeros(n, np.int64)
        return [np.ascontiguousarray(e[r == b]) for b in range(g.insert_blocks)]
    if how == "uneven":
        cuts = np.sort(rng.integers(0, n + 1, size=5))
        parts = np.split(e, cuts)
        parts.insert(2, e[:0])                       # an empty block in the middle
        return [np.ascontiguousarray(p) for p in parts]
    raise ValueError(how)


def run_insert_case(g: Geometry, table_image, blocks):
    """Returns (gpu_table, oracle_table, info)."""
    og = ogeom(g)
    expect = table_image.copy()
    O.insert(og, expect, blocks)
    idx = HashIndex(g)
    idx.load_table(table_image)
    idx.insert([dev(b.reshape(-1, 3)) for b in blocks])
    info = idx.last_update_info()
    got = idx.table_numpy()
    return got, expect, info


def first_diff(a, b):
    d = np.argwhere(a != b)
    if d.size == 0:
        return None
    r = int(d[0][0])
    return {"bucket": r, "gpu": a[r].tolist(), "oracle": b[r].tolist(), "ndiff_buckets": int(len(np.unique(d[:, 0])))}
