"""megakv_amd -- MI355X-native batched GPU hash index behind Mega-KV's
libgpuhash C ABI.

The product is the HIP library built from megakv_amd/csrc (libmegakv_hip.so /
lib/libgpuhash.a).  This package is the thin host-side mirror of the
reference's operator interface used by the tests and the benchmark; PyTorch is
only used to own device memory and streams.
"""
from .lib import load_library, LibraryMissing  # noqa: F401
from .index import (  # noqa: F401
    Geometry, HashIndex, SCHEME_CUCKOO, SCHEME_2CHOICE,
    gpu_hash_search, gpu_hash_insert, gpu_hash_delete,
)
