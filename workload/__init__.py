"""Synthetic request streams (numpy), binding workload/workload.c.

Benchmark and test input only: not part of the product package (megakv_amd).

Restates the reference's LOCAL_TEST workload definition (src/mega_recv.c:633-768,
src/zipf.h) -- see workload.c for the line-by-line citations and SURVEY 8(d):
  K-ref keying  sig = k, hash = ((k & 0xff) << 24 << (8-IBLOCK_P)) | k
  K-mix keying  splitmix64(k) split hi/lo (build-added, exercises alternates)
  GET keys      Zipf(theta, n = P-2, seed 21) rank + 1;  theta = 0 is uniform
  SET keys      S-new: fresh sequential ids from P;  S-upd: drawn like GETs
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
WORKLOAD_PATH = os.path.join(_HERE, "_lib", "libmkv_workload.so")


class LibraryMissing(RuntimeError):
    pass


KEYING_REF = 0
KEYING_MIX = 1
BIN_REGION = 0   # libgpuhash/test/insert_test.c:117-124
BIN_APP = 1      # src/mega_recv.c:476-477


class _Zipf(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("theta", ctypes.c_double), ("alpha", ctypes.c_double),
                ("thres", ctypes.c_double), ("last_n", ctypes.c_uint64), ("dbl_n", ctypes.c_double),
                ("zetan", ctypes.c_double), ("eta", ctypes.c_double), ("rand_state", ctypes.c_uint64)]


class _Synth(ctypes.Structure):
    _fields_ = [("zipf", _Zipf), ("set_key", ctypes.c_uint32), ("num_get", ctypes.c_int),
                ("num_set", ctypes.c_int), ("def_get", ctypes.c_int), ("def_set", ctypes.c_int),
                ("key_len", ctypes.c_int), ("value_len", ctypes.c_int), ("bits_insert_buf", ctypes.c_int)]


_lib = None


def _L():
    global _lib
    if _lib is None:
        if not os.path.exists(WORKLOAD_PATH):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = ctypes.CDLL(WORKLOAD_PATH)
        P = ctypes.c_void_p
        L.mkv_zipf_init.restype = ctypes.c_int
        L.mkv_zipf_init.argtypes = [ctypes.POINTER(_Zipf), ctypes.c_uint64, ctypes.c_double, ctypes.c_uint64]
        L.mkv_zipf_next.restype = ctypes.c_uint64
        L.mkv_zipf_next.argtypes = [ctypes.POINTER(_Zipf)]
        L.mkv_zipf_keys.restype = None
        L.mkv_zipf_keys.argtypes = [ctypes.POINTER(_Zipf), P, ctypes.c_int64]
        L.mkv_make_selem.restype = None
        L.mkv_make_selem.argtypes = [P, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, P]
        L.mkv_make_ielem.restype = None
        L.mkv_make_ielem.argtypes = [P, P, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, P]
        L.mkv_synth_init.restype = ctypes.c_int
        L.mkv_synth_init.argtypes = [ctypes.POINTER(_Synth), ctypes.c_uint64, ctypes.c_double, ctypes.c_uint64,
                                     ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int]
        L.mkv_synth_packets.restype = ctypes.c_int64
        L.mkv_synth_packets.argtypes = [ctypes.POINTER(_Synth), P, ctypes.c_int64, ctypes.c_int64, P, P, P,
                                        ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.mkv_bin_inserts.restype = None
        L.mkv_bin_inserts.argtypes = [P, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, P, P]
        _lib = L
    return _lib


def preload_count(mem_p, load_factor=0.2):
    """P = (uint32)(LOAD_FACTOR * 2^(MEM_P-3))  (src/mega_recv.c:659, src/macros.h:62)"""
    return int(np.uint32(load_factor * ((1 << mem_p) // 8)))


class ZipfKeys:
    """GET key stream of the reference: mehcached_zipf_init(n = P-2, theta, 21)."""

    def __init__(self, n, theta, seed=21):
        self.s = _Zipf()
        if _L().mkv_zipf_init(ctypes.byref(self.s), n, float(theta), seed) != 0:
            raise ValueError("zipf parameters rejected (n > 0, theta in [0,1) or -1 or >= 40)")

    def next_rank(self):
        return int(_L().mkv_zipf_next(ctypes.byref(self.s)))

    def keys(self, count):
        out = np.empty(count, dtype=np.uint32)
        _L().mkv_zipf_keys(ctypes.byref(self.s), out.ctypes.data_as(ctypes.c_void_p), count)
        return out

    def params(self):
        return {"zetan": self.s.zetan, "eta": self.s.eta, "thres": self.s.thres}


class PacketSynth:
    """Request packets as the reference's LOCAL_TEST receiver builds them
    (src/mega_recv.c:633-768): NUM_DEFINED_GET Zipf GETs alternating with NUM_DEFINED_SET
    fresh-key SETs, packed into 1514-byte frames; only the UDP payloads are produced."""

    STRIDE = 1472

    def __init__(self, zipf_n, theta, seed, first_set_key, num_get, num_set, key_len=8, value_len=8, iblock_p=3):
        self.s = _Synth()
        if _L().mkv_synth_init(ctypes.byref(self.s), zipf_n, float(theta), seed, first_set_key, num_get, num_set,
                               key_len, value_len, iblock_p) != 0:
            raise ValueError("packet synthesis parameters rejected")

    def packets(self, npackets):
        """-> payloads uint8 [npackets, 1472], lengths int32 [npackets], GET ids, SET ids (job order)"""
        pay = np.zeros((npackets, self.STRIDE), dtype=np.uint8)
        lens = np.zeros(npackets, dtype=np.int32)
        gk = np.zeros(npackets * 368, dtype=np.uint32)
        sk = np.zeros(npackets * 368, dtype=np.uint32)
        ng, ns = ctypes.c_int64(0), ctypes.c_int64(0)
        P = ctypes.c_void_p
        rc = _L().mkv_synth_packets(ctypes.byref(self.s), pay.ctypes.data_as(P), self.STRIDE, npackets,
                                    lens.ctypes.data_as(P), gk.ctypes.data_as(P), sk.ctypes.data_as(P),
                                    ctypes.byref(ng), ctypes.byref(ns))
        if rc < 0:
            raise RuntimeError("mkv_synth_packets failed")
        return pay, lens, gk[:ng.value].copy(), sk[:ns.value].copy()


def make_selem(keys, keying=KEYING_REF, iblock_p=3):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    out = np.empty((keys.shape[0], 2), dtype=np.uint32)
    _L().mkv_make_selem(keys.ctypes.data_as(ctypes.c_void_p), keys.shape[0], keying, iblock_p,
                        out.ctypes.data_as(ctypes.c_void_p))
    return out


def make_ielem(keys, locs=None, keying=KEYING_REF, iblock_p=3):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    lp = None
    if locs is not None:
        locs = np.ascontiguousarray(locs, dtype=np.uint32)
        lp = locs.ctypes.data_as(ctypes.c_void_p)
    out = np.empty((keys.shape[0], 3), dtype=np.uint32)
    _L().mkv_make_ielem(keys.ctypes.data_as(ctypes.c_void_p), lp, keys.shape[0], keying, iblock_p,
                        out.ctypes.data_as(ctypes.c_void_p))
    return out


def bin_inserts(ielem, mem_p, iblock_p=3, convention=BIN_REGION):
    """Stable split of insert requests into 2^iblock_p blocks; returns list of arrays."""
    ielem = np.ascontiguousarray(ielem, dtype=np.uint32).reshape(-1, 3)
    n = ielem.shape[0]
    out = np.empty_like(ielem)
    counts = np.zeros(1 << iblock_p, dtype=np.int64)
    _L().mkv_bin_inserts(ielem.ctypes.data_as(ctypes.c_void_p), n, mem_p, iblock_p, convention,
                         out.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p))
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [out[offs[b]:offs[b + 1]] for b in range(1 << iblock_p)]
