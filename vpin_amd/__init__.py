"""vpin_amd -- MI355X-native Spartan sat-proof hot path for vt-asaplab/vPIN.

The product is the C-ABI shared library `vpin_amd/lib/libvpin_hip.so` (declared in
include/vpin_hip.h, built from vpin_amd/csrc/ by vpin_amd/build.py).  This Python package is
only a thin ctypes binding over that ABI for the tests and bench.py; it contains no
arithmetic of its own and NO CPU fallback: loading fails loudly when the HIP library is
missing, and every call fails with VPIN_ENODEV when no gfx950 device is usable.
"""
from .capi import (  # noqa: F401
    VpinError, Context, Table, Gens, Comm, dist_plan, gadget_shape, lib, lib_path, KERNEL_CLASSES, exported_symbols, declared_symbols,
)

__all__ = ["VpinError", "Context", "Table", "Gens", "Comm", "dist_plan", "gadget_shape", "lib", "lib_path", "KERNEL_CLASSES",
           "exported_symbols", "declared_symbols"]
