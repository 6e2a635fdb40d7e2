"""Host-side mirror of the reference's libgpuhash interface.

`gpu_hash_search / gpu_hash_insert / gpu_hash_delete` take the same arguments,
in the same order and with the same meaning as the C entry points
(reference libgpuhash/libgpuhash.h:29-51), with torch CUDA tensors standing in
for the device pointers; they call the C ABI of the HIP library and nothing
else.  `HashIndex` is the run-time-geometry convenience used by the tests and
the benchmark (table ownership, request staging), mirroring what the
reference's scheduler does around the three calls
(src/mega_scheduler.c:273-280, 393-504).
"""
import ctypes
from dataclasses import dataclass

import torch

from .lib import MkvGeom, MkvUpdateInfo, load_library

SCHEME_CUCKOO = 0
SCHEME_2CHOICE = 1

MKV_SEARCH_KEEP_MISS = 0
MKV_SEARCH_ZERO_MISS = 1
MKV_CYCLE_MEMSET = 0x100
MKV_CYCLE_NO_OVERLAP = 0x200
MKV_CYCLE_SIDE_STREAM = 0x400


@dataclass(frozen=True)
class Geometry:
    """Run-time form of the gpu_hash.h macros (reference gpu_hash.h:54-75)."""
    mem_p: int = 30
    scheme: int = SCHEME_CUCKOO
    iblock_p: int = 3
    max_cuckoo: int = 5

    @property
    def nbuckets(self):
        return 1 << (self.mem_p - 6)

    @property
    def ht_size(self):
        return 1 << self.mem_p

    @property
    def hash_mask(self):
        return self.nbuckets - 1

    @property
    def block_hash_mask(self):
        return (1 << (self.mem_p - 6 - self.iblock_p)) - 1

    @property
    def insert_blocks(self):
        return 1 << self.iblock_p

    def c(self):
        return MkvGeom(self.mem_p, self.iblock_p, self.scheme, self.max_cuckoo)


def _stream_ptr(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    if isinstance(stream, int):
        return ctypes.c_void_p(stream)
    return ctypes.c_void_p(stream.cuda_stream)


def _dev_ptr(t):
    # (pinned host memory is mapped into the device's address space at the same address: the
    # kernels can read requests from it and write results to it over PCIe, see tools/zero_copy_probe.py)
    assert (t.is_cuda or t.is_pinned()) and t.is_contiguous(), "contiguous CUDA (or pinned host) tensor expected"
    return ctypes.c_void_p(t.data_ptr())


# ---- the reference's three entry points, by name ---------------------------
def gpu_hash_search(in_, out, hash_table, num_elem, num_thread, threads_per_blk, stream=None):
    """reference gpu_hash.cu:482-518; geometry/scheme are the library's baked-in
    MEM_P / HASH_* (like the reference).  `out` must be zeroed by the caller."""
    load_library().gpu_hash_search(_dev_ptr(in_), _dev_ptr(out), _dev_ptr(hash_table),
                                   int(num_elem), int(num_thread), int(threads_per_blk),
                                   _stream_ptr(stream))


def gpu_hash_insert(hash_table, blk_input, blk_elem_num, num_blks, stream=None):
    """reference gpu_hash.cu:521-556.  blk_input: device int64 tensor holding
    num_blks device pointers; blk_elem_num: device int32 tensor of counts."""
    load_library().gpu_hash_insert(_dev_ptr(hash_table), _dev_ptr(blk_input),
                                   _dev_ptr(blk_elem_num), int(num_blks), _stream_ptr(stream))


def gpu_hash_delete(in_, hash_table, num_elem, num_thread, threads_per_blk, stream=None):
    """reference gpu_hash.cu:558-593."""
    load_library().gpu_hash_delete(_dev_ptr(in_), _dev_ptr(hash_table), int(num_elem),
                                   int(num_thread), int(threads_per_blk), _stream_ptr(stream))


class MkvError(RuntimeError):
    pass


class SearchTimer:
    """HIP event pairs recorded by the library around the search launch of a cycle."""

    def __init__(self, slots):
        self.lib = load_library()
        self.slots = slots
        self.h = self.lib.mkv_timer_create(slots)
        if not self.h:
            raise MkvError("mkv_timer_create failed")

    def elapsed_ms(self):
        out = []
        v = ctypes.c_float()
        for i in range(self.slots):
            _check(self.lib.mkv_timer_elapsed_ms(ctypes.c_void_p(self.h), i, ctypes.byref(v)), "mkv_timer_elapsed_ms")
            out.append(v.value)
        return out

    def close(self):
        if self.h:
            self.lib.mkv_timer_destroy(ctypes.c_void_p(self.h))
            self.h = None


def _check(rc, what):
    if rc != 0:
        raise MkvError(f"{what} failed with code {rc}")


class HashIndex:
    """A device-resident table plus the three batched operations.

    Requests and results are torch tensors on the index's device:
      search:  selem int32/uint32 [n,2] (sig, hash)      -> out [n,2] (loc or 0)
      insert:  list of ielem [n_b,3] blocks (sig, hash, loc), canonical order =
               block order then row order
      delete:  delem [n,3]
    """

    def __init__(self, geom: Geometry = Geometry(), device=None, lib=None):
        # (lib: tests pass the diagnostics build, load_library(diagnostics=True))
        self.lib = lib if lib is not None else load_library()   # raises if the HIP library is absent
        if not torch.cuda.is_available():
            raise MkvError("no GPU visible: the hash index has no CPU path")
        self.geom = geom
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._g = geom.c()
        # caller-owned, zero-initialised table (src/mega_scheduler.c:273-274)
        self.table = torch.zeros(geom.ht_size // 4, dtype=torch.int32, device=self.device)
        self._keep = []

    # -- search ----------------------------------------------------------------
    def search(self, selem, out=None, zero_miss=True, stream=None):
        n = selem.shape[0]
        assert selem.shape[1] == 2 and selem.element_size() == 4
        if out is None:
            out = torch.empty((n, 2), dtype=torch.int32, device=self.device)
            if not zero_miss:
                out.zero_()
        flags = MKV_SEARCH_ZERO_MISS if zero_miss else MKV_SEARCH_KEEP_MISS
        _check(self.lib.mkv_search(ctypes.byref(self._g), _dev_ptr(selem), _dev_ptr(out),
                                   _dev_ptr(self.table), n, flags, _stream_ptr(stream)), "mkv_search")
        return out

    # -- insert ----------------------------------------------------------------
    def stage_blocks(self, blocks):
        """Builds the (blk_input, blk_elem_num) device arrays the ABI takes from a
        list of [n_b,3] device tensors.  Returns (ptr_tensor, count_tensor, keepalive)."""
        blocks = [b.contiguous() for b in blocks]
        ptrs = torch.tensor([b.data_ptr() for b in blocks], dtype=torch.int64).to(self.device)
        cnts = torch.tensor([b.shape[0] for b in blocks], dtype=torch.int32).to(self.device)
        return ptrs, cnts, blocks

    def insert(self, blocks, stream=None, hint=None):
        ptrs, cnts, keep = self.stage_blocks(blocks)
        if hint is None:
            hint = sum(b.shape[0] for b in keep)      # the host staged them: it knows the size
        self.insert_staged(ptrs, cnts, len(keep), stream, hint)
        self._keep = [ptrs, cnts, keep]   # alive until the next call (async launch)

    def insert_staged(self, ptrs, cnts, num_blks, stream=None, hint=0):
        _check(self.lib.mkv_insert_hinted(ctypes.byref(self._g), _dev_ptr(self.table), _dev_ptr(ptrs),
                                          _dev_ptr(cnts), num_blks, int(hint), _stream_ptr(stream)), "mkv_insert")

    def cycle(self, selem, out, ptrs=None, cnts=None, num_blks=0, delem=None, memset=False, overlap="fused",
              stream=None, timer=None, slot=0, insert_hint=0):
        """search -> delete -> insert of one scheduler cycle in one library call.
        overlap: "fused" (mark pass inside the search launch), "side" (side stream), "none"."""
        if overlap is True:
            overlap = "fused"
        if overlap is False:
            overlap = "none"
        flags = (MKV_CYCLE_MEMSET if memset else 0) | {"fused": 0, "side": MKV_CYCLE_SIDE_STREAM,
                                                        "none": MKV_CYCLE_NO_OVERLAP}[overlap]
        null = ctypes.c_void_p(0)
        _check(self.lib.mkv_cycle(
            ctypes.byref(self._g), _dev_ptr(self.table),
            _dev_ptr(selem) if selem is not None else null, _dev_ptr(out) if out is not None else null,
            selem.shape[0] if selem is not None else 0,
            _dev_ptr(delem) if delem is not None else null, delem.shape[0] if delem is not None else 0,
            _dev_ptr(ptrs) if num_blks else null, _dev_ptr(cnts) if num_blks else null, num_blks,
            int(insert_hint), flags, _stream_ptr(stream), ctypes.c_void_p(timer) if timer else null, slot), "mkv_cycle")

    # -- delete ----------------------------------------------------------------
    def delete(self, delem, stream=None):
        assert delem.shape[1] == 3 and delem.element_size() == 4
        self._keep = [delem]
        _check(self.lib.mkv_delete(ctypes.byref(self._g), _dev_ptr(delem), _dev_ptr(self.table),
                                   delem.shape[0], _stream_ptr(stream)), "mkv_delete")

    # -- misc --------------------------------------------------------------------
    def reserve(self, max_requests, stream=None):
        _check(self.lib.mkv_reserve(ctypes.byref(self._g), int(max_requests), _stream_ptr(stream)),
               "mkv_reserve")

    def sorted_stats(self, stream=None, cycle_insert=False):
        """big insert passes of the device's workspace so far: {passes, fallbacks, left_over (last pass)}"""
        out = (ctypes.c_uint32 * 4)()
        _check(self.lib.mkv_sorted_stats(_stream_ptr(stream), 1 if cycle_insert else 0, ctypes.byref(out)), "mkv_sorted_stats")
        return {"passes": int(out[0]), "fallbacks": int(out[1]), "left_over": int(out[2])}

    def last_update_info(self, stream=None, cycle_insert=False):
        info = MkvUpdateInfo()
        _check(self.lib.mkv_last_update_info_ex(_stream_ptr(stream), 1 if cycle_insert else 0,
                                                ctypes.byref(info)), "mkv_last_update_info")
        return {k: int(getattr(info, k)) for k, _ in info._fields_}

    def tail_phases_us(self, stream=None, cycle_insert=False):
        """diagnostics: time of the last call's tail kernels per phase (roll-back, gather,
        exact replay, retag, in-group rounds, serial), microseconds"""
        out = (ctypes.c_float * 16)()
        if not hasattr(self.lib, "mkv_debug_tail_phases"):
            raise MkvError("tail_phases_us: diagnostics build only (HashIndex(lib=load_library(diagnostics=True)))")
        _check(self.lib.mkv_debug_tail_phases(_stream_ptr(stream), 1 if cycle_insert else 0, out),
               "mkv_debug_tail_phases")
        return [round(float(x), 1) for x in out]

    def table_numpy(self):
        """Table image as uint32 [nbuckets,16] on the host (sig[8] | loc[8])."""
        return self.table.cpu().numpy().view("uint32").reshape(-1, 16)

    def load_table(self, image):
        """image: numpy uint32 [nbuckets,16]."""
        t = torch.from_numpy(image.reshape(-1).view("int32"))
        self.table.copy_(t)
