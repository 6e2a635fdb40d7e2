"""Two independent readings of the reference's insert kernels against each other (CPU only).

oracle/gpuhash_oracle.c states the kernels' canonical serial outcome in collapsed form and is
what every parity test compares the HIP library with.  oracle/lane_emulator.c walks down
libgpuhash/gpu_hash.cu:77-229 and :231-433 lane by lane for one 8-lane group at a time (real
ballot / __ffs bit arithmetic including the group's bit_move offset, the major-location
rotation, the write / __syncthreads / re-read claim loops left in).  Neither is the reference
-- its CUDA source cannot be built here -- but the reference's own tests pin neither slot
choice nor the eviction chain, so two readings that agree over a large fuzz
campaign are the best evidence there is.  A disagreeing seed would be committed here."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-s", "-C", HERE, "_build/liblane_check.so"])
    lib = ctypes.CDLL(os.path.join(HERE, "_build", "liblane_check.so"))
    lib.lane_fuzz.restype = ctypes.c_int64
    lib.lane_fuzz.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint64),
                              ctypes.POINTER(ctypes.c_uint64 * 4)]
    lib.lane_insert.restype = None
    lib.lane_insert.argtypes = [ctypes.POINTER(O.Geom), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return lib


def lane_insert(L, g, table, blocks):
    blocks = [np.ascontiguousarray(b, dtype=np.uint32).reshape(-1, 3) for b in blocks]
    ptrs = (ctypes.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
    cnts = (ctypes.c_int * len(blocks))(*[b.shape[0] for b in blocks])
    L.lane_insert(ctypes.byref(g), table.ctypes.data_as(ctypes.c_void_p), ptrs, cnts, len(blocks))


def test_two_readings_agree_over_a_fuzz_campaign(L):
    """300 000 sequences: tables of 4..256 buckets at every fill, up to 512 requests each from
    small signature spaces (duplicates, full buckets, chains, dropped victims), signature 0
    with and without loc 0, alternate == primary, victims whose alternate is the bucket they
    were evicted from, 1-4 blocks, both schemes, MAX_CUCKOO_NUM 0..6."""
    bad = ctypes.c_uint64(0)
    totals = (ctypes.c_uint64 * 4)()
    n = L.lane_fuzz(1, 300000, ctypes.byref(bad), ctypes.byref(totals))
    assert n == 0, f"the two readings differ on {n} sequences, first seed {bad.value}"
    requests, evictions, dropped, skipped = (int(x) for x in totals)
    assert requests > 5e7 and evictions > 1e6 and dropped > 1e5 and skipped > 1e5, list(totals)


def test_emulator_reproduces_the_survey_recorded_behaviours(L):
    """What SURVEY 8(c) recorded from executing the reference kernels, through the lane-level
    reading: slot order 6, 7, 0, 1 for sig & 7 == 6; victim 0x1003 leaves bucket 0x206 for
    0x1006 = inserter_hash ^ victim_sig (not its own alternate); two-choice overflow keeps the
    old loc under the new signature."""
    g = O.geom(mem_p=30)
    t = O.new_table(g)
    reqs = np.array([[0x1006 + 8 * i, 77, 500 + i] for i in range(4)], np.uint32)      # sig & 7 == 6
    lane_insert(L, g, t, [reqs])
    assert [int(np.flatnonzero(t[77, :8] == r[0])[0]) for r in reqs] == [6, 7, 0, 1]
    t = O.new_table(g)
    t[5, :8] = np.arange(0x5000, 0x5008)
    t[5, 8:] = np.arange(100, 108)
    t[0x206, :8] = np.arange(0x1000, 0x1008)
    t[0x206, 8:] = np.arange(200, 208)
    lane_insert(L, g, t, [np.array([[0x203, 5, 777]], np.uint32)])
    assert t[0x206, 3] == 0x203 and t[0x206, 11] == 777
    assert t[0x1006, 3] == 0x1003 and t[0x1006, 11] == 203 and not t[0x1205].any()
    g2 = O.geom(mem_p=30, scheme=O.SCHEME_2CHOICE)
    t = O.new_table(g2)
    t[5, :8] = np.arange(0x5000, 0x5008)
    t[5, 8:] = np.arange(100, 108)
    t[0x206, :8] = np.arange(0x1000, 0x1008)
    t[0x206, 8:] = np.arange(200, 208)
    lane_insert(L, g2, t, [np.array([[0x203, 5, 777]], np.uint32)])
    assert t[0x206, 3] == 0x203 and t[0x206, 11] == 203                     # stale loc (ref :197-209)


@pytest.mark.parametrize("scheme", [O.SCHEME_CUCKOO, O.SCHEME_2CHOICE])
def test_emulator_equals_oracle_on_a_larger_table(L, scheme):
    g = O.geom(mem_p=18, scheme=scheme)
    rng = np.random.default_rng(3 + scheme)
    e = np.empty((60000, 3), np.uint32)
    e[:, 0] = rng.integers(1, 1 << 20, size=len(e))
    e[:, 1] = rng.integers(0, 2**32, size=len(e), dtype=np.uint64)
    e[:, 2] = rng.integers(1, 2**32, size=len(e), dtype=np.uint64)
    a, b = O.new_table(g), O.new_table(g)
    parts = [e[:20000], e[20000:20001], e[20001:]]
    lane_insert(L, g, a, parts)
    st = O.insert(g, b, parts)
    assert np.array_equal(a, b)
    assert scheme == O.SCHEME_2CHOICE or st["evictions"] > 1000
