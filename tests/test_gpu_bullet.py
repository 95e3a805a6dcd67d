"""Kernel-level parity of the bullet reduction on the device (bullet.hip, msm.hip bullet_step_kernel) against
BulletReductionProof::prove restated literally (Spartan/src/nizk/bullet.rs:32-132): the model below FOLDS the generator
vector every round (G' = u^-1 G_L + u G_R with the oracle's group arithmetic, oracle/group.c) and the vectors with Python
integers mod q, while the device never folds G (fixed-base MSMs with scalars a'[.] s_j formed on the fly) -- so equal
L_k, R_k, c_L, c_R, x_hat, a_hat and g_hat for given challenges check the reformulation itself, the one-side-per-
workgroup layout (n >= 32), the split tree (n < 32), the double-buffered folds and the mailbox.  Both device paths:
the fused one-launch rounds (R up to 32768) and the three-launch rounds behind them (VPIN_BULLET_CLASSIC, odd sizes).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import pymodel as M

pytestmark = pytest.mark.gpu
Q = M.Q
NB = 1024


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gens(ctx):
    xyzt, og = O.gens_stream_xyzt(NB, b"test bullet gens")
    g = ctx.gens_create(xyzt)
    yield g, og
    g.free()


@pytest.fixture(scope="module")
def big_gens(ctx):
    """the shared table of b"gens_r1cs_eval" (32770 generators, the two-segment layout of tests/test_gpu_msm_wide.py)"""
    xyzt, og = O.gens_stream_xyzt(32770, b"gens_r1cs_eval")
    g = ctx.gens_shared("gens_r1cs_eval", xyzt, 80 if ctx.device_total_bytes() >= (200 << 30) else 24)
    return g, og


def msm(points, ints):
    """sum_i ints[i] * points[i] through the oracle (Ge)"""
    L = O.lib()
    n = len(ints)
    pts = (O.Ge * n)(*points)
    r = O.Ge()
    L.ge_msm(C.byref(r), O.ptr(np.ascontiguousarray(M.ints_to_table(ints))), pts, n)
    return r


def compress(p):
    out = (C.c_uint8 * 32)()
    O.lib().ge_compress(out, C.byref(p))
    return bytes(out)


def model(a, b, G, us):
    """bullet.rs:52-131 without Q, H and the transcript: per round (c_L, c_R, L, R), then (x_hat, a_hat, g_hat)"""
    rounds = []
    n = len(a)
    for u in us:
        n //= 2
        ui = pow(u, Q - 2, Q)
        aL, aR, bL, bR, GL, GR = a[:n], a[n:], b[:n], b[n:], G[:n], G[n:]
        cL = sum(x * y for x, y in zip(aL, bR)) % Q
        cR = sum(x * y for x, y in zip(aR, bL)) % Q
        rounds.append((cL, cR, compress(msm(GR, aL)), compress(msm(GL, aR))))
        a = [(aL[i] * u + ui * aR[i]) % Q for i in range(n)]
        b = [(bL[i] * ui + u * bR[i]) % Q for i in range(n)]
        G = [msm([GL[i], GR[i]], [ui, u]) for i in range(n)]
    return rounds, a[0], b[0], compress(G[0])


def device(ctx, g, a, b, us, classic):
    import vpin_amd
    L = vpin_amd.lib()
    R, k = len(a), len(us)
    ta, tb, tu = M.ints_to_table(a), M.ints_to_table(b), M.ints_to_table(us)
    cLR = np.zeros((k, 2, 4), dtype=np.uint64)
    LR = np.zeros((k, 2, 32), dtype=np.uint8)
    fin = np.zeros((2, 4), dtype=np.uint64)
    gh = np.zeros(32, dtype=np.uint8)
    L.vpin_bullet_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.vpin_bullet_reduce(ctx.h, g.h, ta.ctypes.data_as(C.c_void_p), tb.ctypes.data_as(C.c_void_p), R,
                              tu.ctypes.data_as(C.c_void_p), classic, cLR.ctypes.data_as(C.c_void_p),
                              LR.ctypes.data_as(C.c_void_p), fin.ctypes.data_as(C.c_void_p), gh.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return cLR, LR, fin, bytes(gh)


def vectors(rng, R, case):
    if case == "random":
        a = [int.from_bytes(rng.bytes(40), "little") % Q for _ in range(R)]
    elif case == "sparse":  # zeros, ones, q-1: zero scalars skip their table walks, whole workgroups may hold the identity
        pool = [0, 0, 0, 1, Q - 1, 2, int.from_bytes(rng.bytes(40), "little") % Q]
        a = [pool[int(i)] for i in rng.integers(0, len(pool), size=R)]
    else:  # left half zero: every L is the identity in round 0
        a = [0] * (R // 2) + [int.from_bytes(rng.bytes(40), "little") % Q for _ in range(R // 2)]
    b = [int.from_bytes(rng.bytes(40), "little") % Q for _ in range(R)]
    return a, b


@pytest.mark.parametrize("R,case,classic", [(64, "random", 0), (256, "random", 0), (256, "sparse", 0), (256, "half", 0),
                                            (1024, "random", 0), (32, "random", 0), (256, "random", 1), (16, "sparse", 1),
                                            (1024, "sparse", 1)])
def test_bullet_rounds_match_the_folding_model(ctx, gens, R, case, classic):
    g, og = gens
    rng = np.random.default_rng(R * 7 + len(case) + classic)
    a, b = vectors(rng, R, case)
    k = R.bit_length() - 1
    us = [int.from_bytes(rng.bytes(40), "little") % (Q - 1) + 1 for _ in range(k)]
    want_rounds, x_hat, a_hat, g_hat = model(a, b, [og[i] for i in range(R)], us)
    cLR, LR, fin, gh = device(ctx, g, a, b, us, classic)
    for j, (cL, cR, Lc, Rc) in enumerate(want_rounds):
        assert M.table_to_ints(cLR[j]) == [cL, cR], f"inner products of round {j}"
        assert bytes(LR[j, 0]) == Lc, f"L of round {j} (n = {R >> (j + 1)})"
        assert bytes(LR[j, 1]) == Rc, f"R of round {j} (n = {R >> (j + 1)})"
    assert M.table_to_ints(fin) == [x_hat, a_hat]
    assert gh == g_hat


@pytest.mark.parametrize("R,case,classic", [(8192, "random", 0), (32768, "sparse", 0), (8192, "sparse", 1)])
def test_long_rows_match_the_folding_model(ctx, big_gens, R, case, classic):
    """rows of more than 4096 scalars (instances of 2^24 constraints and more): fused rounds whose workgroup points are summed
    per 128 on the device (R = 32768 also walks the narrow-window table segment), and the three-launch rounds with
    msm_wide_kernel<4>"""
    g, og = big_gens
    rng = np.random.default_rng(R + len(case) + classic)
    a, b = vectors(rng, R, case)
    k = R.bit_length() - 1
    us = [int.from_bytes(rng.bytes(40), "little") % (Q - 1) + 1 for _ in range(k)]
    want_rounds, x_hat, a_hat, g_hat = model(a, b, [og[i] for i in range(R)], us)
    cLR, LR, fin, gh = device(ctx, g, a, b, us, classic)
    for j, (cL, cR, Lc, Rc) in enumerate(want_rounds):
        assert M.table_to_ints(cLR[j]) == [cL, cR], f"inner products of round {j}"
        assert bytes(LR[j, 0]) == Lc and bytes(LR[j, 1]) == Rc, f"L / R of round {j} (n = {R >> (j + 1)})"
    assert M.table_to_ints(fin) == [x_hat, a_hat]
    assert gh == g_hat


def test_fused_rounds_are_refused_where_they_do_not_exist(ctx, gens):
    """R = 16 is not a multiple of the 32 generators a workgroup covers: classic == 0 must say so, not fall back silently"""
    import vpin_amd
    g, _ = gens
    L = vpin_amd.lib()
    t = M.ints_to_table([1] * 16)
    u = M.ints_to_table([3] * 4)
    out = np.zeros(4096, dtype=np.uint8)
    L.vpin_bullet_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    p = out.ctypes.data_as(C.c_void_p)
    assert L.vpin_bullet_reduce(ctx.h, g.h, t.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), 16,
                                u.ctypes.data_as(C.c_void_p), 0, p, p, p, p) != 0
