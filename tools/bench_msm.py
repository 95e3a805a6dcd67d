#!/usr/bin/env python3
"""bench_msm.py -- row-commitment MSM rate on full-width random scalars, for a few generator-set
sizes (the window width c follows from the table budget).  Prints table adds/s from HIP events.
usage: python tools/bench_msm.py [--nb 4098,16386,32770] [--log-entries 24]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

Q = 2**252 + 27742317777372353535851937790883648493


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", default="4098,16386,32770")
    ap.add_argument("--log-entries", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import vpin_amd
    from vpin_amd import capi
    ctx = vpin_amd.Context(0)
    rng = np.random.default_rng(1)
    n = 1 << args.log_entries
    # random 256-bit values with the top nibble cleared are < q with overwhelming probability; as
    # Montgomery images they are just some other uniformly random scalars
    Z = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    Z[:, 3] &= np.uint64((1 << 59) - 1)
    tZ = ctx.upload(Z)
    for nb in [int(x) for x in args.nb.split(",")]:
        R = nb - 2
        L = n // R
        xyzt = np.zeros((nb, 128), dtype=np.uint8)
        capi.lib().vpin_host_gens_derive(b"gens_r1cs_eval", nb, xyzt.ctypes.data_as(capi.C.c_void_p))
        t0 = time.perf_counter()
        g = ctx.gens_create(xyzt)
        t_build = time.perf_counter() - t0
        blinds = np.zeros((L, 4), dtype=np.uint64)
        sub = ctx.wrap(tZ.device_ptr(), L * R) if L * R != n else tZ
        ctx.hyrax_commit(g, sub, blinds, R + 1)
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(args.reps):
            ctx.hyrax_commit(g, sub, blinds, R + 1)
        st = ctx.prof_read()["msm"]
        ctx.prof_enable(False)
        ms = st["ms"] / st["launches"]
        c = next(c for c in range(12, 6, -1) if nb * ((254 + c - 1) // c) * (1 << (c - 1)) * capi.lib().vpin_gens_entry_bytes() <= (int(os.environ.get('VPIN_GENS_BUDGET_GB', '24')) << 30))
        W = (254 + c - 1) // c
        print(f"nb={nb} c={c} W={W} table={nb * W * (1 << (c - 1)) * capi.lib().vpin_gens_entry_bytes() / 2**30:.1f} GiB build={t_build:.2f}s  "
              f"L={L} R={R}: {ms:.2f} ms  {L * R / ms / 1e6:.3f} G scalars/s  {L * R * W / ms / 1e6:.2f} G adds/s", flush=True)
        g.free()
    ctx.close()


if __name__ == "__main__":
    main()
