"""Product-round kernel time against the number of pairs (development aid): python tools/ubench_rounds.py [ncirc]
HIP-event time per launch of prod_round_kernel<true, true, *> through vpin_spark_batched_round, garbage tables."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpin_amd  # noqa: E402

vp = C.c_void_p
ncirc = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = vpin_amd.Context(0)
if True:
    L = vpin_amd.lib()
    L.vpin_spark_batched_round.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, vp, C.c_int,
                                           vp, vp, vp, C.c_int, vp]
    top = 21
    h = 4 << top  # level-0 half length: a bound round of 2^top pairs has len = h
    n = 2 * h
    forest = ctx.alloc(1 << (ncirc * 2 * n - 1).bit_length())  # table lengths are powers of two
    E = ctx.alloc(h)
    r = np.array([[0x1234567, 0x89abcdef, 0x1111, 0x0123456789abcde]], dtype=np.uint64)
    out = np.zeros((ncirc, 3, 4), dtype=np.uint64)
    ctx.prof_enable(1)
    for lg in range(top, 8, -1):
        length = 4 << lg  # bound round: pairs = length / 4
        for rep in range(3):
            ctx.prof_reset()
            reps = 5
            for _ in range(reps):
                rc = L.vpin_spark_batched_round(ctx.h, forest.h, n, ncirc, 0, length, E.h, 0, r.ctypes.data_as(vp), 1, None, None, None, 0,
                                                out.ctypes.data_as(vp))
                assert rc == 0, rc
            st = ctx.prof_read()
        k = st.get("spark_round") or st.get("spark_round_big")
        us = 1e3 * k["ms"] / k["launches"]
        pairs = 1 << lg
        print(f"pairs 2^{lg:2d} x {ncirc}: {us:9.1f} us/launch   {ncirc * pairs / us / 1e3:7.2f} G pair-evals/s", flush=True)
