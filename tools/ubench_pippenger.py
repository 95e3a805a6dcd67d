#!/usr/bin/env python3
"""tools/ubench_pippenger.py -- the row commitment two ways on the same resident scalars and generators (run on the GPU box):

  walk       vpin_hyrax_commit: the production window-table walk (msm.hip; a c = 12 table over the generators)
  pippenger  vpin_hyrax_commit_pippenger: bucket accumulation staged in LDS (msm_pip.hip), c = 9 .. 12

for the two row shapes of the LeNet trace's largest commitments (R = 4096 and R = 16384 generators) and two kinds of scalars
(full-width, as the SPARK polynomials' evaluations of eq; witness-like: 35 % zeros, bits, small values).  Per run: kernel time
from the context's HIP-event profile, point additions counted on the device (table additions / bucket additions), the bytes
compared with the walk's, sclk and socket power sampled from hwmon.  One text table to stdout, one JSON line at the end.

  python3 tools/ubench_pippenger.py [--rows-small 4096] [--rows-large 2048] [--loops 3]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from bench_common import PowerSampler  # noqa: E402


def scalars(kind, n, rng):
    z = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    z[:, 3] &= np.uint64((1 << 60) - 1)   # below 2^252 < q: a canonical Montgomery image of some full-width scalar
    if kind == "witness":
        # Montgomery images of 0, 1 and 16-bit values: 0 stays 0; the others are full-width images whose VALUES are small --
        # the kernels work on the value, so build the images of small values: v * R mod q
        Q = 2**252 + 27742317777372353535851937790883648493
        Rm = (1 << 256) % Q
        k = rng.random(n)
        small = np.where(k < 0.35, 0, np.where(k < 0.45, 1, rng.integers(0, 2**16, size=n)))
        pick = k < 0.5
        tab = np.zeros((1 << 16, 4), dtype=np.uint64)
        for v in range(1 << 16):
            m = (v * Rm) % Q
            tab[v] = [(m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)]
        z[pick] = tab[small[pick]]
    return z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-small", type=int, default=4096)
    ap.add_argument("--rows-large", type=int, default=2048)
    ap.add_argument("--loops", type=int, default=3)
    ap.add_argument("--shapes", default="4096,16384")
    ap.add_argument("--narrow", default="40,20,10,7,4,1", help="table budgets in GB for the narrow-window walks at R = 16384 ('' = none)")
    a = ap.parse_args()
    os.environ.setdefault("VPIN_GENS_BUDGET_GB", "96")   # the production width: 12-bit windows over 16386 generators are 71 GB
    import ctypes as C
    import vpin_amd
    ctx = vpin_amd.Context(0)
    cus, clk = ctx.device_props()
    print(f"device: {cus} CUs, nominal {clk / 1e3:.0f} MHz", flush=True)
    results = []
    for R in [int(x) for x in a.shapes.split(",")]:
        L = a.rows_small if R <= 4096 else a.rows_large
        t0 = time.perf_counter()
        stream = hashlib.shake_256(b"ubench_pippenger").digest(64 * (R + 2))
        xyzt = ctx.gens_map_stream(stream)
        g = ctx.gens_create(xyzt)
        lay6 = (C.c_size_t * 6)()
        vpin_amd.lib().vpin_gens_layout.argtypes = [C.c_void_p, C.c_void_p]
        vpin_amd.lib().vpin_gens_layout(g.h, lay6)
        lay = f"c = {lay6[0]}, W = {lay6[1]}, first segment {lay6[2]} bases" + (f", then c = {lay6[3]}, W = {lay6[4]}" if lay6[3] else "")
        print(f"R = {R}: {R + 2} generators, window table built in {time.perf_counter() - t0:.1f} s (layout {lay}); {L} rows = {L * R / 2**20:.0f} Mi scalars", flush=True)
        zero_bl = np.zeros((L, 4), dtype=np.uint64)
        for kind in ("full-width", "witness"):
            rng = np.random.default_rng(R + len(kind))
            dZ = ctx.upload(scalars(kind, L * R, rng))
            variants = [("walk", None)] + [("pippenger", c) for c in (9, 10, 11, 12)]
            # the walk over NARROWER tables (what a part without 71 GB to spare would build): budgets in GB
            narrow = {}
            if R >= 16384 and a.narrow:
                for gb in [int(x) for x in a.narrow.split(",")]:
                    gn = ctx.gens_shared(f"ubench_pippenger_{R}_{gb}", xyzt, gb)
                    vpin_amd.lib().vpin_gens_layout(gn.h, lay6)
                    narrow[f"walk, {gb} GB budget: c = {lay6[0]}, W = {lay6[1]}"] = gn
                variants += [(nm, gn) for nm, gn in narrow.items()]
            ref = None
            for name, c in variants:
                if c is None:
                    run = lambda: ctx.hyrax_commit(g, dZ, zero_bl, R + 1)
                elif isinstance(c, int):
                    run = lambda c=c: ctx.hyrax_commit_pippenger(g, dZ, zero_bl, R + 1, c_bits=c)
                else:
                    run = lambda gn=c: ctx.hyrax_commit(gn, dZ, zero_bl, R + 1)
                out = run()   # warm (and the bytes)
                dig = hashlib.sha256(out.tobytes()).hexdigest()
                if ref is None:
                    ref = dig
                ctx.prof_enable(2)
                ctx.prof_reset()
                ps = PowerSampler(0, interval_s=0.02)
                ps.start()
                t1 = time.perf_counter()
                for _ in range(a.loops):
                    run()
                t2 = time.perf_counter()
                pw = ps.stop(t1 + 0.5 * (t2 - t1), t2) or {}
                st = ctx.prof_read()
                ctx.prof_enable(False)
                ms = (st.get("msm", {}).get("ms", 0.0)) / a.loops
                adds = (st.get("msm_rows", {}).get("units", 0.0)) / a.loops
                r = dict(R=R, rows=L, scalars=kind, variant=f"pippenger c={c}" if isinstance(c, int) else name, kernel_ms=round(ms, 3),
                         wall_ms=round((t2 - t1) * 1e3 / a.loops, 3), counted_adds=adds, G_adds_s=round(adds / ms / 1e6, 3) if ms else None,
                         G_scalars_s=round(L * R / ms / 1e6, 4) if ms else None, adds_per_scalar=round(adds / (L * R), 3),
                         bytes_equal_walk=(dig == ref), sclk_mhz=pw.get("sclk_mhz_median"), watts=pw.get("watts_median"))
                results.append(r)
                print(f"  {kind:10s} {r['variant']:38s} {r['kernel_ms']:10.2f} ms  {r['G_scalars_s']} G scalars/s  {r['adds_per_scalar']:6.2f} counted adds/scalar "
                      f" {r['G_adds_s']} G adds/s  sclk {r['sclk_mhz']} W {r['watts']}  bytes {'ok' if r['bytes_equal_walk'] else 'DIFF'}", flush=True)
            dZ.free()
        g.free()
    ctx.close()
    print("JSON " + json.dumps({"device_cus": cus, "loops": a.loops,
                                "note": "counted adds: table additions (walk) / bucket additions = non-zero digits (pippenger; its running-sum, "
                                        "suffix-scan, tree and Horner additions are extra and not counted)", "runs": results}))
    return 0 if all(r["bytes_equal_walk"] for r in results) else 2


if __name__ == "__main__":
    sys.exit(main())
