"""CPU-side check that the HIP library loads and exports every symbol the
headers under include/ declare (no compute, no GPU needed), and that a plain-C
host compiles against the headers with gcc and links against lib/libgpuhash.a."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(diagnostics=False):
    """what the headers declare: outside their `#ifdef MKV_DIAGNOSTICS` sections (the product
    library), or inside them (what the diagnostics build adds)"""
    names = set()
    for h in ("libgpuhash.h", "megakv_hip.h", "cuda_runtime.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        diag = "".join(re.findall(r"#ifdef MKV_DIAGNOSTICS\n(.*?)#endif", txt, flags=re.S))
        txt = diag if diagnostics else re.sub(r"#ifdef MKV_DIAGNOSTICS\n.*?#endif", "", txt, flags=re.S)
        txt = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", txt)
        for m in re.finditer(r"\b(gpu_[a-z_]+|mkv_[a-z_]+|cuda[A-Z][A-Za-z]+)\s*\(", txt):
            names.add(m.group(1))
    names -= {"mkv_geom_s", "mkv_update_info_s"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from megakv_amd.lib import ABI_SYMBOLS, LIB_PATH, load_library
    if not os.path.exists(LIB_PATH):
        pytest.fail(f"{LIB_PATH} missing: run __graft_entry__.build()")
    L = load_library()
    decl = declared_functions()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
        assert name in ABI_SYMBOLS, f"{name} not covered by the loader's symbol list"


def test_diagnostics_stay_out_of_the_product_library():
    """mkv_debug_* entry points exist in the diagnostics build (libmegakv_hip_diag.so) only; neither
    the product shared object nor the static archive a C host links has any."""
    from megakv_amd.lib import DIAG_LIB_PATH, DIAG_SYMBOLS, LIB_PATH, load_library
    decl = declared_functions(diagnostics=True)
    assert sorted(decl) == sorted(DIAG_SYMBOLS)
    D = load_library(diagnostics=True)
    for name in decl:
        assert hasattr(D, name), f"{name} missing from {DIAG_LIB_PATH}"
    for lib in (LIB_PATH, os.path.join(ROOT, "lib", "libgpuhash.a")):
        syms = subprocess.check_output(["nm", "--defined-only", lib], text=True, stderr=subprocess.DEVNULL)
        assert "mkv_debug" not in syms, lib


def test_layouts_match_reference_header():
    """sizeof/offsetof of the ABI structs, as the reference header yields with gcc
    (SURVEY section 0): bucket_t 64 B (sig@0, loc@32), selem_t 8 B, ielem_t 12 B."""
    src = r'''
#include <stddef.h>
#include <stdio.h>
#include "gpu_hash.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %llu %u %u %d\n", sizeof(bucket_t), offsetof(bucket_t, sig),
    offsetof(bucket_t, loc), sizeof(selem_t), offsetof(selem_t, hash), sizeof(ielem_t),
    offsetof(ielem_t, loc), sizeof(delem_t), (unsigned long long)HT_SIZE, HASH_MASK, BLOCK_HASH_MASK, INSERT_BLOCK);
  return 0; }'''
    exe = "/tmp/mkv_layout_test"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe],
                   input=src, text=True, check=True)
    out = subprocess.check_output([exe], text=True).split()
    assert out == ["64", "0", "32", "8", "4", "12", "8", "12", str(1 << 30), str(0xFFFFFF), str(0x1FFFFF), "8"]


def test_plain_c_host_links_against_static_archive():
    lib = os.path.join(ROOT, "lib", "libgpuhash.a")
    if not os.path.exists(lib):
        pytest.fail("lib/libgpuhash.a missing: run __graft_entry__.build()")
    drv = os.path.join(ROOT, "examples", "scheduler_cycle.c")
    exe = "/tmp/mkv_scheduler_cycle"
    subprocess.run(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), drv, "-o", exe,
                    "-L", os.path.join(ROOT, "lib"), "-lgpuhash", "-L/opt/rocm/lib", "-lamdhip64",
                    "-lstdc++", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    assert os.path.exists(exe)
