"""ctypes binding of oracle/gpuhash_oracle.c (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- see gpuhash_oracle.c for what is restated and how
it is pinned.  Arrays use the reference layouts (libgpuhash/gpu_hash.h:79-104):
  table  uint32[nbuckets, 16]  (sig[8] then loc[8])
  selem  uint32[n, 2]          (sig, hash)
  ielem  uint32[n, 3]          (sig, hash, loc)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgpuhash_oracle.so")

SCHEME_CUCKOO = 0
SCHEME_2CHOICE = 1


class Geom(ctypes.Structure):
    _fields_ = [("mem_p", ctypes.c_uint32), ("iblock_p", ctypes.c_uint32),
                ("scheme", ctypes.c_uint32), ("max_cuckoo", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    _fields_ = [("elems", ctypes.c_uint64), ("buckets_visited", ctypes.c_uint64),
                ("dword_stores", ctypes.c_uint64), ("evictions", ctypes.c_uint64),
                ("dropped", ctypes.c_uint64), ("skipped_zero", ctypes.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("gpuhash_oracle.c", "gpuhash_oracle.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None
_native = False


def use_native():
    """cpu_baseline only: switch to the same source built -march=native ON THIS MACHINE (into
    the temporary directory -- such an object must not travel with the tree).  Returns whether
    the native flavour is in use; on any failure the portable build stays."""
    global _lib, _SO, _native
    if _native:
        return True
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"mkv_oracle_native_{os.getuid()}")
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native", f"NATIVE_OUT={out}"])
        so = os.path.join(out, "libgpuhash_oracle_native.so")
        ctypes.CDLL(so)
    except Exception:                                             # noqa: BLE001
        return False
    _SO, _lib, _native = so, None, True
    return True


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        P = ctypes.c_void_p
        L.orc_primary.restype = ctypes.c_uint32
        L.orc_primary.argtypes = [ctypes.POINTER(Geom), ctypes.c_uint32]
        L.orc_alternate.restype = ctypes.c_uint32
        L.orc_alternate.argtypes = [ctypes.POINTER(Geom), ctypes.c_uint32, ctypes.c_uint32]
        L.orc_search.restype = None
        L.orc_search.argtypes = [ctypes.POINTER(Geom), P, P, P, ctypes.c_int64]
        L.orc_search_mt.restype = None
        L.orc_search_mt.argtypes = [ctypes.POINTER(Geom), P, P, P, ctypes.c_int64, ctypes.c_int]
        L.orc_insert.restype = None
        L.orc_insert.argtypes = [ctypes.POINTER(Geom), P, P, P, ctypes.c_int, ctypes.POINTER(Stats)]
        L.orc_delete.restype = None
        L.orc_delete.argtypes = [ctypes.POINTER(Geom), P, P, ctypes.c_int64, ctypes.POINTER(Stats)]
        _lib = L
    return _lib


def geom(mem_p=30, scheme=SCHEME_CUCKOO, iblock_p=3, max_cuckoo=5):
    return Geom(mem_p, iblock_p, scheme, max_cuckoo)


def nbuckets(g):
    return 1 << (g.mem_p - 6)


def new_table(g):
    """Zeroed table; np.zeros is lazily committed so MEM_P=30 costs nothing
    until buckets are touched."""
    return np.zeros((nbuckets(g), 16), dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(a, cols):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"], "uint32 C-contiguous expected"
    assert a.ndim == 2 and a.shape[1] == cols, f"shape [n,{cols}] expected, got {a.shape}"


def primary(g, hash_):
    return int(lib().orc_primary(ctypes.byref(g), ctypes.c_uint32(int(hash_))))


def alternate(g, hash_, sig):
    return int(lib().orc_alternate(ctypes.byref(g), ctypes.c_uint32(int(hash_)),
                                   ctypes.c_uint32(int(sig))))


def search(g, table, selem, out=None, threads=1):
    _check(table, 16)
    _check(selem, 2)
    n = selem.shape[0]
    if out is None:
        out = np.zeros((n, 2), dtype=np.uint32)
    _check(out, 2)
    if threads > 1:
        lib().orc_search_mt(ctypes.byref(g), _ptr(table), _ptr(selem), _ptr(out), n, threads)
    else:
        lib().orc_search(ctypes.byref(g), _ptr(table), _ptr(selem), _ptr(out), n)
    return out


def insert(g, table, blocks):
    """blocks: list of uint32[n_b, 3] arrays, one per insert block, processed in
    list order then index order (the canonical order)."""
    _check(table, 16)
    blocks = [np.ascontiguousarray(b, dtype=np.uint32).reshape(-1, 3) for b in blocks]
    nb = len(blocks)
    ptrs = (ctypes.c_void_p * max(nb, 1))(*[b.ctypes.data for b in blocks])
    cnts = (ctypes.c_int * max(nb, 1))(*[b.shape[0] for b in blocks])
    st = Stats()
    lib().orc_insert(ctypes.byref(g), _ptr(table), ptrs, cnts, nb, ctypes.byref(st))
    return st.as_dict()


def insert_mt(g, table, blocks):
    """insert() with one host thread per block (blocks binned by closed region); falls back to
    insert() when they are not.  For the CPU baseline."""
    _check(table, 16)
    blocks = [np.ascontiguousarray(b, dtype=np.uint32).reshape(-1, 3) for b in blocks]
    nb = len(blocks)
    ptrs = (ctypes.c_void_p * max(nb, 1))(*[b.ctypes.data for b in blocks])
    cnts = (ctypes.c_int * max(nb, 1))(*[b.shape[0] for b in blocks])
    L = lib()
    L.orc_insert_mt.restype = ctypes.c_int
    L.orc_insert_mt.argtypes = [ctypes.POINTER(Geom), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    if not L.orc_insert_mt(ctypes.byref(g), _ptr(table), ptrs, cnts, nb):
        insert(g, table, blocks)
        return False
    return True


def delete(g, table, delem):
    _check(table, 16)
    _check(delem, 3)
    st = Stats()
    lib().orc_delete(ctypes.byref(g), _ptr(table), _ptr(delem), delem.shape[0], ctypes.byref(st))
    return st.as_dict()


def parse_packets(payloads, lens, key_len, value_len, fold=False):
    """payloads: uint8 [npackets, stride]; lens: int32 [npackets].  Returns (gets [n,2],
    sets [n,3], bad_packets, bad_jobs) as the reference's receiver would batch them."""
    L = lib()
    L.orc_parse_packet.restype = ctypes.c_int
    L.orc_parse_packet.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                   ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    payloads = np.ascontiguousarray(payloads, dtype=np.uint8)
    npk, stride = payloads.shape
    gets = np.zeros((npk * 512, 2), np.uint32)
    sets = np.zeros((npk * 512, 3), np.uint32)
    ng, ns, nb = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    bad_packets = 0
    for p in range(npk):
        n = int(min(int(lens[p]), stride, 2048))
        rc = L.orc_parse_packet(payloads[p].ctypes.data, max(n, 0), p, key_len, value_len, 1 if fold else 0,
                                gets.ctypes.data, ctypes.byref(ng), sets.ctypes.data, ctypes.byref(ns),
                                ctypes.byref(nb))
        bad_packets += 1 if rc != 0 else 0
    return gets[:ng.value].copy(), sets[:ns.value].copy(), bad_packets, nb.value


def resolve_verify(pairs, keys, key_len, arena, item_stride, nkey_off=10, key_off=24, bytes_mode=False):
    """orc_resolve_verify: the sender's pick + KEY_MATCH compare (src/mega_send.c:408-414, 543-553).
    pairs [n,2] uint32, keys [n,key_stride] uint8, arena [items,item_stride] uint8.
    Returns (loc [n] uint32, hits, misses, false positives)."""
    L = lib()
    L.orc_resolve_verify.restype = None
    L.orc_resolve_verify.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    keys = np.ascontiguousarray(keys, dtype=np.uint8)
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    n = pairs.shape[0]
    out = np.zeros(n, np.uint32)
    tot = np.zeros(3, np.uint64)
    L.orc_resolve_verify(pairs.ctypes.data, n, keys.ctypes.data, keys.shape[1] if n else key_len, key_len,
                         arena.ctypes.data, item_stride, arena.shape[0], nkey_off, key_off, 1 if bytes_mode else 0,
                         out.ctypes.data, tot.ctypes.data)
    return out, int(tot[0]), int(tot[1]), int(tot[2])
