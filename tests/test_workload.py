"""Workload generator (workload/workload.c) against the reference's own
Zipf generator: committed known answers (tests/golden/zipf_kat.json, produced
by tests/golden/make_golden.py from /root/reference/src/zipf.h) and, where the
reference-built helper oracle/_ref/ref_zipf is present, a live cross-check.
Also BASELINE config 1 (LOCAL_TEST-shaped, CPU only) end to end on the oracle."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import workload as W
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "zipf_kat.json")))


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"n{c['n']}-t{c['theta']}")
def test_zipf_matches_reference_known_answers(case):
    z = W.ZipfKeys(case["n"], case["theta"], case["seed"])
    ranks = z.keys(case["count"]).astype(np.uint64) - 1      # keys are rank + 1
    assert ranks[:64].tolist() == case["first"]
    digest = hashlib.sha256(ranks.astype("<u8").tobytes()).hexdigest()
    assert digest == case["sha256_le64"]
    p = z.params()
    assert p["zetan"] == case["zetan"] and p["eta"] == case["eta"] and p["thres"] == case["thres"]


def test_zipf_live_against_reference_binary():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_zipf")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_zipf not built (reference tree absent)")
    n, theta, seed, count = 123457, 0.9, 5, 20000
    txt = subprocess.check_output([exe, str(n), repr(theta), str(seed), str(count)], text=True).split("\n")
    ref = np.array([int(x) for x in txt[:count]], dtype=np.uint64)
    z = W.ZipfKeys(n, theta, seed)
    assert np.array_equal(z.keys(count).astype(np.uint64) - 1, ref)


def test_zipf_skew():
    z = W.ZipfKeys(26843543, 0.99, 21)
    k = z.keys(200000)
    assert abs((k == 1).mean() - 0.052) < 0.004         # P(rank 0) ~ 0.0520


def test_kref_keying():
    """key k -> sig = k, hash = ((k & 7) << 29) | k  (src/mega_recv.c:700-704,361-362)"""
    k = np.array([1, 2, 7, 8, 255, 256, 0x12345678, 26843544], dtype=np.uint32)
    s = W.make_selem(k, W.KEYING_REF)
    assert np.array_equal(s[:, 0], k)
    assert np.array_equal(s[:, 1], ((k & 7) << 29) | k)
    e = W.make_ielem(k, None, W.KEYING_REF)
    assert np.array_equal(e[:, :2], s) and np.array_equal(e[:, 2], k)


def test_kmix_keying_never_zero_sig_and_is_stable():
    k = np.arange(1, 200001, dtype=np.uint32)
    s = W.make_selem(k, W.KEYING_MIX)
    assert (s[:, 0] != 0).all()
    assert np.array_equal(s, W.make_selem(k, W.KEYING_MIX))
    # hashes spread over regions
    assert len(np.unique((s[:, 1] & 0xFFFFFF) >> 21)) == 8


@pytest.mark.parametrize("conv", [W.BIN_REGION, W.BIN_APP])
def test_bin_inserts_is_stable_partition(conv):
    r = np.random.default_rng(4)
    e = r.integers(0, 2**32, size=(5000, 3), dtype=np.uint64).astype(np.uint32)
    blocks = W.bin_inserts(e, 30, 3, conv)
    assert sum(b.shape[0] for b in blocks) == 5000
    for b, blk in enumerate(blocks):
        key = ((blk[:, 1] & 0xFFFFFF) >> 21) if conv == W.BIN_REGION else (blk[:, 1] >> 29)
        assert (key == b).all()
        # stable: rows appear in their original relative order
        idx = [np.nonzero((e == row).all(axis=1))[0][0] for row in blk[:50]]
        assert idx == sorted(idx)


def test_config1_local_test_shape_on_cpu():
    """BASELINE config 1: MEM_P=25, two-choice, K-ref keys, preload P-1 SETs
    (LF 0.2) in batches of <= 8 x 4096 binned the app's way, then uniform GET
    batches of 32767: every GET hits in its primary bucket with loc == key."""
    mem_p = 25
    g = O.geom(mem_p=mem_p, scheme=O.SCHEME_2CHOICE)
    t = O.new_table(g)
    P = W.preload_count(mem_p)
    assert P == 838860
    keys = np.arange(1, P, dtype=np.uint32)
    for lo in range(0, keys.shape[0], 8 * 4095):
        e = W.make_ielem(keys[lo:lo + 8 * 4095], None, W.KEYING_REF)
        blocks = W.bin_inserts(e, mem_p, 3, W.BIN_APP)
        assert max(b.shape[0] for b in blocks) <= 4096
        O.insert(g, t, blocks)
    z = W.ZipfKeys(P - 2, 0.0, 21)
    blk_mask = (1 << (mem_p - 9)) - 1
    for _ in range(4):
        k = z.keys(32767)
        assert k.max() < P
        out = O.search(g, t, W.make_selem(k, W.KEYING_REF))
        assert np.array_equal(out[:, 0], k)                      # 100 % hits, loc = key
        # alternate = region base bucket: only keys whose low bits are zero live there
        assert np.array_equal(out[:, 1], np.where((k & blk_mask) == 0, k, 0))
