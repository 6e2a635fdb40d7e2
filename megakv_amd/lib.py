"""ctypes loader for the HIP library.  There is no fallback: if the shared
object is missing or does not export the C ABI, importing code fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libmegakv_hip.so")
# the same sources built with -DMKV_DIAGNOSTICS (make -C megakv_amd/csrc DIAG=1): the mkv_debug_* entry points and
# the path-forcing tunings of tests and tools; never what bench.py or a host links
DIAG_LIB_PATH = os.path.join(_HERE, "_lib", "libmegakv_hip_diag.so")


class LibraryMissing(RuntimeError):
    pass


class MkvGeom(ctypes.Structure):
    _fields_ = [("mem_p", ctypes.c_uint32), ("iblock_p", ctypes.c_uint32),
                ("scheme", ctypes.c_uint32), ("max_cuckoo", ctypes.c_uint32)]


class MkvWorker(ctypes.Structure):
    _fields_ = [("search_in", ctypes.c_void_p), ("search_out", ctypes.c_void_p), ("num_search", ctypes.c_int64),
                ("delete_in", ctypes.c_void_p), ("num_delete", ctypes.c_int64),
                ("blk_input", ctypes.c_void_p), ("blk_elem_num", ctypes.c_void_p), ("num_blks", ctypes.c_int)]


class MkvUpdateInfo(ctypes.Structure):
    _fields_ = [("total", ctypes.c_uint32), ("round1", ctypes.c_uint32), ("round2", ctypes.c_uint32),
                ("tail_parallel", ctypes.c_uint32), ("tail_serial", ctypes.c_uint32),
                ("undone", ctypes.c_uint32), ("first_hard", ctypes.c_uint32), ("chains", ctypes.c_uint32),
                ("bar_chain", ctypes.c_uint32), ("bar_class", ctypes.c_uint32), ("tail_rounds", ctypes.c_uint32),
                ("helped", ctypes.c_uint32), ("losers", ctypes.c_uint32)]


# every symbol include/libgpuhash.h, include/megakv_hip.h and
# include/cuda_runtime.h declare
ABI_SYMBOLS = [
    "gpu_hash_search", "gpu_hash_insert", "gpu_hash_delete", "gpu_delete_insert",
    "mkv_search", "mkv_insert", "mkv_delete", "mkv_reserve", "mkv_release_all",
    "mkv_last_update_info", "mkv_route_requests", "mkv_unroute_results", "mkv_version",
    "mkv_set_tuning", "mkv_cycle", "mkv_cycle_workers", "mkv_insert_hinted", "mkv_pack_slot_words", "mkv_pack_scratch_bytes", "mkv_pack_requests",
    "mkv_search_slots", "mkv_unpack_results", "mkv_pack_one_launch", "mkv_pack_meta_words", "mkv_pack_requests_meta", "mkv_unpack_results_tiled", "mkv_slots_cycle", "mkv_derive_requests", "mkv_resolve_results", "mkv_resolve_verify", "mkv_last_update_info_ex", "mkv_timer_create", "mkv_timer_elapsed_ms", "mkv_timer_destroy",
    "mkv_shard_create", "mkv_shard_unique_id", "mkv_shard_connect", "mkv_shard_run", "mkv_shard_overflowed",
    "mkv_shard_error", "mkv_shard_destroy", "mkv_shard_trace", "mkv_shard_phases", "mkv_parse_scratch_bytes", "mkv_parse_packets",
    "mkv_set_defer", "mkv_flush", "mkv_forget_stream", "mkv_dropped_batches", "mkv_sorted_stats", "mkv_workspace_bytes", "mkv_pipe_create", "mkv_pipe_run", "mkv_pipe_error", "mkv_pipe_destroy", "mkv_host_alloc", "mkv_host_free",
    "cudaMalloc", "cudaFree", "cudaHostAlloc", "cudaFreeHost", "cudaMemset",
    "cudaMemsetAsync", "cudaMemcpy", "cudaMemcpyAsync", "cudaStreamCreate",
    "cudaStreamDestroy", "cudaStreamSynchronize", "cudaDeviceSynchronize",
    "cudaGetLastError", "cudaGetErrorString", "cudaMemGetInfo",
]

DIAG_SYMBOLS = ["mkv_debug_tail_phases", "mkv_debug_stamp_resets", "mkv_debug_random_read", "mkv_debug_request",
                "mkv_debug_wide_round"]

_lib = None
_diag_lib = None


def load_library(diagnostics=False):
    """The product library; diagnostics=True: the diagnostics build (a second, independent copy
    of the library in this process -- its workspaces and tunings are its own)."""
    global _lib, _diag_lib
    if diagnostics:
        if _diag_lib is None:
            _diag_lib = _load(DIAG_LIB_PATH, True)
        return _diag_lib
    if _lib is None:
        _lib = _load(LIB_PATH, False)
    return _lib


def _load(LIB_PATH, diagnostics):
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C megakv_amd/csrc{' DIAG=1' if diagnostics else ''}`.  There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    P, I64, U32, INT = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int
    G = ctypes.POINTER(MkvGeom)
    # (MKV_LIB_OLD_BUILD=1, tools/ab_lines.sh only: an earlier round's build of the library stands in the product's
    # place for an A/B -- entry points it does not have yet are simply not there)
    old_build = bool(os.environ.get("MKV_LIB_OLD_BUILD"))
    for name in ABI_SYMBOLS + (DIAG_SYMBOLS if diagnostics else []):
        if not hasattr(L, name):
            if old_build:
                class _Absent:
                    restype = argtypes = None
                setattr(L, name, _Absent())
                continue
            raise LibraryMissing(f"{LIB_PATH} does not export {name}")
    L.gpu_hash_search.restype = None
    L.gpu_hash_search.argtypes = [P, P, P, INT, INT, INT, P]
    L.gpu_hash_insert.restype = None
    L.gpu_hash_insert.argtypes = [P, P, P, INT, P]
    L.gpu_hash_delete.restype = None
    L.gpu_hash_delete.argtypes = [P, P, INT, INT, INT, P]
    L.mkv_search.restype = INT
    L.mkv_search.argtypes = [G, P, P, P, I64, U32, P]
    L.mkv_insert.restype = INT
    L.mkv_insert.argtypes = [G, P, P, P, INT, P]
    L.mkv_delete.restype = INT
    L.mkv_delete.argtypes = [G, P, P, I64, P]
    L.mkv_cycle.restype = INT
    L.mkv_cycle.argtypes = [G, P, P, P, I64, P, I64, P, P, INT, I64, U32, P, P, INT]
    L.mkv_cycle_workers.restype = INT
    L.mkv_cycle_workers.argtypes = [G, P, ctypes.POINTER(MkvWorker), INT, P]
    L.mkv_insert_hinted.restype = INT
    L.mkv_insert_hinted.argtypes = [G, P, P, P, INT, ctypes.c_longlong, P]
    L.cudaMemsetAsync.restype = INT
    L.cudaMemsetAsync.argtypes = [P, INT, ctypes.c_size_t, P]
    L.mkv_set_defer.restype = None
    L.mkv_set_defer.argtypes = [INT]
    L.mkv_flush.restype = INT
    L.mkv_flush.argtypes = []
    L.mkv_dropped_batches.restype = ctypes.c_ulonglong
    L.mkv_dropped_batches.argtypes = []
    L.mkv_sorted_stats.restype = INT
    L.mkv_sorted_stats.argtypes = [P, INT, ctypes.POINTER(U32 * 4)]
    L.mkv_workspace_bytes.restype = ctypes.c_longlong
    L.mkv_workspace_bytes.argtypes = []
    L.mkv_timer_create.restype = P
    L.mkv_timer_create.argtypes = [INT]
    L.mkv_timer_elapsed_ms.restype = INT
    L.mkv_timer_elapsed_ms.argtypes = [P, INT, ctypes.POINTER(ctypes.c_float)]
    L.mkv_timer_destroy.restype = None
    L.mkv_timer_destroy.argtypes = [P]
    L.mkv_reserve.restype = INT
    L.mkv_reserve.argtypes = [G, I64, P]
    L.mkv_release_all.restype = None
    L.mkv_last_update_info.restype = INT
    L.mkv_last_update_info.argtypes = [P, ctypes.POINTER(MkvUpdateInfo)]
    L.mkv_last_update_info_ex.restype = INT
    L.mkv_last_update_info_ex.argtypes = [P, INT, ctypes.POINTER(MkvUpdateInfo)]
    if diagnostics:
        L.mkv_debug_random_read.restype = ctypes.c_longlong
        L.mkv_debug_random_read.argtypes = [P, ctypes.c_longlong, INT, ctypes.c_longlong, ctypes.c_longlong, P, P]
        L.mkv_debug_wide_round.restype = INT
        L.mkv_debug_wide_round.argtypes = [INT]
        L.mkv_debug_stamp_resets.restype = INT
        L.mkv_debug_stamp_resets.argtypes = [P, INT]
        L.mkv_debug_tail_phases.restype = INT
        L.mkv_debug_tail_phases.argtypes = [P, INT, ctypes.POINTER(ctypes.c_float)]
    L.mkv_route_requests.restype = INT
    L.mkv_route_requests.argtypes = [P, INT, I64, U32, U32, INT, P, P, P, P, I64, P]
    L.mkv_pack_slot_words.restype = I64
    L.mkv_pack_slot_words.argtypes = [I64, I64]
    L.mkv_pack_scratch_bytes.restype = I64
    L.mkv_pack_scratch_bytes.argtypes = [I64, I64, INT]
    L.mkv_pack_requests.restype = INT
    L.mkv_pack_requests.argtypes = [P, I64, P, I64, U32, U32, INT, P, I64, I64, P, P, I64, P, P]
    L.mkv_search_slots.restype = INT
    L.mkv_search_slots.argtypes = [G, P, INT, I64, I64, P, P, P, P]
    L.mkv_parse_scratch_bytes.restype = I64
    L.mkv_parse_scratch_bytes.argtypes = [I64]
    L.mkv_parse_packets.restype = INT
    L.mkv_parse_packets.argtypes = [P, I64, P, I64, INT, INT, INT, P, P, P, I64, I64, P, P, I64, P]
    L.mkv_shard_create.restype = INT
    L.mkv_shard_create.argtypes = [ctypes.POINTER(P), G, INT, INT, P, I64, I64, I64, I64, ctypes.c_char_p]
    L.mkv_shard_unique_id.restype = INT
    L.mkv_shard_unique_id.argtypes = [P, P]
    L.mkv_shard_connect.restype = INT
    L.mkv_shard_connect.argtypes = [P, P]
    L.mkv_shard_run.restype = INT
    L.mkv_shard_run.argtypes = [P, INT, P, P, P, P, P, P, P, INT]
    L.mkv_shard_overflowed.restype = INT
    L.mkv_shard_overflowed.argtypes = [P, ctypes.POINTER(INT)]
    L.mkv_shard_trace.restype = None
    L.mkv_shard_trace.argtypes = [INT]
    L.mkv_shard_phases.restype = INT
    L.mkv_shard_phases.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(INT)]
    L.mkv_shard_error.restype = ctypes.c_char_p
    L.mkv_shard_error.argtypes = [P]
    L.mkv_shard_destroy.restype = None
    L.mkv_shard_destroy.argtypes = [P]
    L.mkv_pipe_create.restype = INT
    L.mkv_pipe_create.argtypes = [ctypes.POINTER(P), G, P, I64, I64, INT, INT]
    L.mkv_pipe_run.restype = INT
    L.mkv_pipe_run.argtypes = [P, INT, P, P, P, P, P, INT, P]
    L.mkv_pipe_error.restype = ctypes.c_char_p
    L.mkv_pipe_error.argtypes = [P]
    L.mkv_pipe_destroy.restype = None
    L.mkv_pipe_destroy.argtypes = [P]
    L.mkv_host_alloc.restype = P
    L.mkv_host_alloc.argtypes = [I64]
    L.mkv_host_free.restype = None
    L.mkv_host_free.argtypes = [P]
    L.mkv_slots_cycle.restype = INT
    L.mkv_slots_cycle.argtypes = [G, P, INT, I64, I64, P, P, P, ctypes.c_longlong, P, P, INT]
    L.mkv_derive_requests.restype = INT
    L.mkv_derive_requests.argtypes = [P, I64, INT, I64, INT, P, P, INT, P]
    L.mkv_resolve_results.restype = INT
    L.mkv_resolve_results.argtypes = [P, I64, P, P, P]
    L.mkv_resolve_verify.restype = INT
    L.mkv_resolve_verify.argtypes = [P, I64, P, I64, INT, P, I64, I64, INT, INT, INT, P, P, P]
    L.mkv_unpack_results.restype = INT
    L.mkv_unpack_results.argtypes = [P, P, I64, P, P]
    L.mkv_unroute_results.restype = INT
    L.mkv_unroute_results.argtypes = [P, P, I64, P, P]
    L.mkv_set_tuning.restype = INT
    L.mkv_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    L.mkv_version.restype = ctypes.c_char_p
    return L
