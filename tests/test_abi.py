"""CPU-side checks of the C ABI: the library builds/loads and exports every declared symbol."""
import ctypes as C
import os
import re

from bbmap_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "bbmap_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bb(?:map|msa|band|idx|pipe|keys)_[a-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    L = C.CDLL(_lib.SO_PATH)
    syms = declared_symbols()
    assert syms, "no symbols parsed from include/bbmap_amd.h"
    for s in syms:
        assert hasattr(L, s), "missing export: " + s
    assert sorted(_lib.EXPORTS) == syms


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.bbmsa_job) == 40
    assert C.sizeof(_lib.bbmsa_result) == 80
    assert _lib.bbmsa_result.iterations.offset == 24
    assert _lib.bbmsa_result.score.offset == 32


def test_abi_version():
    assert _lib.load().bbmap_abi_version() == 6
