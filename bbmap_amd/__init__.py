"""bbmap_amd -- MI355X (gfx950) native implementation of BBMap's seed-and-extend hot path.

The package holds the HIP kernels + C ABI (csrc/, libbbmap_amd.so) and thin Python bindings used by
tests and bench.py.  All compute runs in the HIP library; there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
