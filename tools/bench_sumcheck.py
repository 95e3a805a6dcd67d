"""Kernel-level timing of the sum-check round kernels (development tool; bench.py is the
contract benchmark).  Usage: python tools/bench_sumcheck.py [log2_len ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpin_amd  # noqa: E402


def fast_rand(rng, n, zero_tail=0.0):
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    if zero_tail:
        a[int(n * (1 - zero_tail)):] = 0
    return a


def run(ctx, ell, k, fused, reps=3):
    rng = np.random.default_rng(ell)
    n = 1 << ell
    host = [fast_rand(rng, n) for _ in range(k)]
    r = fast_rand(rng, 1)[0]
    best = None
    for rep in range(reps):
        dev = [ctx.upload(t) for t in host]
        ctx.prof_reset()
        ctx.prof_enable(True)
        t0 = time.perf_counter()
        if k == 4:
            ctx.sc_cubic_round(*dev)
        else:
            ctx.sc_quad_round(*dev)
        while len(dev[0]) >= 4 and fused:
            (ctx.sc_cubic_bind_round(*dev, r) if k == 4 else ctx.sc_quad_bind_round(*dev, r))
        while len(dev[0]) >= 2:
            ctx.sc_bind(dev, r)
            if len(dev[0]) >= 2:
                (ctx.sc_cubic_round(*dev) if k == 4 else ctx.sc_quad_round(*dev))
        wall = time.perf_counter() - t0
        st = ctx.prof_read()
        ctx.prof_enable(False)
        for t in dev:
            t.free()
        kern_ms = sum(v["ms"] for v in st.values())
        if best is None or kern_ms < best["kernel_ms"]:
            best = {"log2_len": ell, "tables": k, "fused": fused, "wall_ms": wall * 1e3, "kernel_ms": kern_ms,
                    "stats": st}
    alg = k * 32.0 * 3 * n  # ~ k*32*(len + len/2) summed over rounds = k*32*1.5*2n
    best["alg_GBps_over_kernel_time"] = alg / (best["kernel_ms"] * 1e-3) / 1e9
    for name, v in best["stats"].items():
        v["GBps"] = v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else None
    return best


if __name__ == "__main__":
    ells = [int(x) for x in sys.argv[1:]] or [16, 20, 22]
    with vpin_amd.Context(0) as ctx:
        for ell in ells:
            for k in (4, 2):
                for fused in (False, True):
                    print(json.dumps(run(ctx, ell, k, fused)))
