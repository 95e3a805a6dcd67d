"""Row-commitment MSM rate against the polynomial size (development aid): python tools/ubench_msm.py
vpin_hyrax_commit of 2^ell full-width random scalars as 2^(ell/2) rows, HIP-event time of msm_rows_kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
import vpin_amd  # noqa: E402

ctx = vpin_amd.Context(0)
NB = 16386
xyzt = np.zeros((NB, 128), dtype=np.uint8)  # MultiCommitGens::new through the product's own host derivation
L = vpin_amd.lib()
L.vpin_host_gens_derive.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
assert L.vpin_host_gens_derive(b"gens_r1cs_eval", NB, xyzt.ctypes.data_as(C.c_void_p)) == 0
g = ctx.gens_shared("gens_r1cs_eval", xyzt, 80 if ctx.device_total_bytes() >= (200 << 30) else 24)
rng = np.random.default_rng(1)
top = 24
host = rng.integers(0, 2**64, size=(1 << top, 4), dtype=np.uint64)
host[:, 3] &= np.uint64((1 << 60) - 1)
ctx.prof_enable(2)
for ell in range(14, top + 1):
    n = 1 << ell
    L = 1 << (ell // 2)
    R = n // L
    sub = ctx.alloc(n)
    sub.write(0, host[:n])
    blinds = host[:L].copy()
    for rep in range(2):
        ctx.prof_reset()
        ctx.hyrax_commit(g, sub, blinds, R)
        st = ctx.prof_read()
    k = st.get("msm_rows") or st.get("msm")
    adds = 22.0 * n  # 12-bit signed windows: ~22 non-zero digits per full-width scalar
    print(f"2^{ell}: {L} rows x {R}: {1e3 * k['ms']:9.1f} us  {adds / (k['ms'] * 1e-3) / 1e9:6.2f} G adds/s  ({k['launches']} launches)", flush=True)
    sub.free()
