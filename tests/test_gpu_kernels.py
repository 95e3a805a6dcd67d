"""Kernel-level GPU parity for the kernels that own the step time, at the sizes where they run multi-block with
many pairs per thread (2^13, 2^20, 2^22 pairs): every variant of the phase-1 round kernel
(sc_cubic3_kernel<BIND,LEAD>), of the SPARK batched round kernel (prod_round_kernel<BIND,LEAD>) and the
dot-product round kernel (dotp_round_kernel<BIND>) against the CPU oracle's round evaluation
(oracle/poly.c: Spartan/src/sumcheck.rs:624-652, dense_mlpoly.rs:229-236), bit-exact, plus the folded tables.
The wide / two-segment / compaction-boundary paths of the MSM are in test_gpu_msm_wide.py.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import pymodel as M

pytestmark = pytest.mark.gpu
Q = M.Q
vp = C.c_void_p


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    L = vpin_amd.lib()
    L.vpin_eq_suffix_tables.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.vpin_sc_cubic3_lead_round.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.vpin_sc_cubic3_round.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.vpin_sc_cubic3_bind_round.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.vpin_spark_batched_round.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, vp, C.c_int,
                                           vp, vp, vp, C.c_int, vp]
    yield c
    c.close()


SPECIAL = [0, 1, Q - 1, Q - 2, (Q - 1) // 2, 2**252 - 1]


def fast_table(rng, n, zero_frac=0.0):
    """n field elements in Montgomery form, (n,4) uint64: uniform below 2^252 (< q, so every row is a valid image),
    a share of zeros, zero tail, and the edge values 0, 1, q-1, ... sprinkled in"""
    t = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    t[:, 3] &= np.uint64((1 << 60) - 1)
    if zero_frac:
        t[rng.random(n) < zero_frac] = 0
        t[n - n // 16:] = 0  # padding tail, as the witness-derived tables have
    if n >= 1024:
        sp = M.ints_to_table(SPECIAL)
        for k in range(4 * len(SPECIAL)):
            t[int(rng.integers(0, n))] = sp[k % len(SPECIAL)]
    return t


def rand_scalar(rng):
    return fast_table(rng, 1)[0].copy()


def to_int(limbs):
    return M.from_mont_limbs(limbs)


def lead_from_evals(e0, e2, e3):
    """t quadratic with t(0), t(2), t(3) known -> (t(0), x^2 coefficient, t(1))"""
    c2 = ((e3 - e0) * pow(3, -1, Q) - (e2 - e0) * pow(2, -1, Q)) % Q
    b = ((e2 - e0 - 4 * c2) * pow(2, -1, Q)) % Q
    return e0, c2, (e0 + b + c2) % Q


def dup(E):
    return np.ascontiguousarray(np.concatenate([E, E]))


# ---- phase 1: sc_cubic3_kernel<BIND, LEAD> / sc_tail3_kernel ------------------------------------------------

@pytest.mark.parametrize("ell,rounds", [(14, 14), (21, 4), (23, 3)])
def test_phase1_lead_rounds_vs_oracle(ctx, ell, rounds):
    """the kernels vpin_sat_prove actually launches: <false,true> for round 0 (t(0), x^2 coefficient, t(1)), then
    <true,true> (constant-fold form of r in LDS, 8 pairs per thread above 2^21 pairs) for the bound rounds;
    ell = 14 walks all the way down through the one-workgroup tail kernels"""
    import vpin_amd
    L = vpin_amd.lib()
    rng = np.random.default_rng(9000 + ell)
    n = 1 << ell
    tau = fast_table(rng, ell)
    host = [fast_table(rng, n, zero_frac=0.3) for _ in range(3)]
    dev = [ctx.upload(t) for t in host]
    pyr = vp()
    assert L.vpin_eq_suffix_tables(ctx.h, tau.ctypes.data_as(vp), ell, C.byref(pyr)) == 0
    zeros = None
    r = None
    for j in range(rounds):
        out = np.zeros((3, 4), dtype=np.uint64)
        if j:
            host = [O.bound_top(t, r) for t in host]
        rc = L.vpin_sc_cubic3_lead_round(ctx.h, pyr, ell, j + 1, dev[0].h, dev[1].h, dev[2].h,
                                         r.ctypes.data_as(vp) if j else None, out.ctypes.data_as(vp))
        assert rc == 0
        E = O.eq_evals(tau[j + 1:]) if j + 1 < ell else M.ints_to_table([1])
        exp = [to_int(x) for x in O.sc_cubic_round(dup(E), *host)]
        t0, c2, t1 = lead_from_evals(*exp)
        got = [to_int(x) for x in out]
        assert got[0] == t0 and got[1] == c2, f"round {j} (pairs {len(E)})"
        if j == 0:
            assert got[2] == t1
        for d, h in zip(dev, host):
            assert len(d) == len(h)
        if j in (1, rounds - 1):
            for d, h in zip(dev, host):
                assert np.array_equal(d.read(), h), f"folded table after round {j}"
        r = rand_scalar(rng)
    L.vpin_table_free(ctx.h, pyr)


@pytest.mark.parametrize("ell", [21, 23])
def test_phase1_three_sum_rounds_at_size(ctx, ell):
    """<false,false> and <true,false> (the fallback when the claim is not the true sum) at multi-block sizes"""
    import vpin_amd
    L = vpin_amd.lib()
    rng = np.random.default_rng(9100 + ell)
    n = 1 << ell
    tau = fast_table(rng, ell)
    host = [fast_table(rng, n, zero_frac=0.3) for _ in range(3)]
    dev = [ctx.upload(t) for t in host]
    pyr = vp()
    assert L.vpin_eq_suffix_tables(ctx.h, tau.ctypes.data_as(vp), ell, C.byref(pyr)) == 0
    r = None
    for j in range(3):
        out = np.zeros((3, 4), dtype=np.uint64)
        if j == 0:
            rc = L.vpin_sc_cubic3_round(ctx.h, pyr, ell, 1, dev[0].h, dev[1].h, dev[2].h, out.ctypes.data_as(vp))
        else:
            host = [O.bound_top(t, r) for t in host]
            rc = L.vpin_sc_cubic3_bind_round(ctx.h, pyr, ell, j + 1, dev[0].h, dev[1].h, dev[2].h, r.ctypes.data_as(vp),
                                             out.ctypes.data_as(vp))
        assert rc == 0
        E = O.eq_evals(tau[j + 1:])
        assert np.array_equal(out, O.sc_cubic_round(dup(E), *host)), f"round {j}"
        r = rand_scalar(rng)
    for d, h in zip(dev, host):
        assert np.array_equal(d.read(), h)
    L.vpin_table_free(ctx.h, pyr)


# ---- SPARK: prod_round_kernel<BIND, LEAD>, dotp_round_kernel<BIND> --------------------------------------------

def next_pow2(x):
    p = 1
    while p < x:
        p *= 2
    return p


def dev_table(ctx, host=None, rows=None):
    """zero-filled device table of rows rounded up to a power of two, `host` copied to its front"""
    rows = next_pow2(rows if rows is not None else host.shape[0])
    t = ctx.alloc(rows)
    if host is not None:
        t.write(0, host)
    return t


def batched_round(ctx, forest_v, n, ncirc, length, E_v, e_off, r, lead, dotp=None, first_fold=0):
    import vpin_amd
    L = vpin_amd.lib()
    out = np.zeros((ncirc + (6 if dotp else 0), 3, 4), dtype=np.uint64)
    d = dotp or (None, None, None)
    rc = L.vpin_spark_batched_round(ctx.h, forest_v.h, n, ncirc, 0, length, E_v.h, e_off,
                                    r.ctypes.data_as(vp) if r is not None else None, int(lead),
                                    d[0].h if dotp else None, d[1].h if dotp else None, d[2].h if dotp else None, first_fold,
                                    out.ctypes.data_as(vp))
    assert rc == 0, rc
    return out


@pytest.mark.parametrize("lg_pairs,ncirc,with_dotp,lead", [(13, 12, True, True), (13, 12, True, False), (20, 12, False, True),
                                                            (20, 4, False, False), (22, 2, False, True), (9, 12, True, True)])
def test_spark_batched_rounds_vs_oracle(ctx, lg_pairs, ncirc, with_dotp, lead):
    """three consecutive rounds of prove_cubic_batched on level 0 of a forest: round 0 unbound (<false,*>), then two
    bound rounds (<true,*>; the dot-product tables fold into scratch first, then in place)"""
    rng = np.random.default_rng(7000 + lg_pairs * 16 + ncirc + 2 * lead)
    h = 2 << lg_pairs  # entries per half: pairs = h/2 in round 0
    n = 2 * h
    zeros = np.zeros((h, 4), dtype=np.uint64)
    A = [fast_table(rng, h, zero_frac=0.1) for _ in range(ncirc)]
    B = [fast_table(rng, h, zero_frac=0.1) for _ in range(ncirc)]
    forest_v = dev_table(ctx, rows=ncirc * 2 * n)
    for t in range(ncirc):
        forest_v.write(t * 2 * n, A[t])
        forest_v.write(t * 2 * n + h, B[t])
    k = lg_pairs + 1  # rounds of this layer; E_j = eq(rand[j+1..]) has h >> (j+1) entries
    rand = fast_table(rng, k)
    Es = [O.eq_evals(rand[j + 1:]) for j in range(3)]
    E_v = dev_table(ctx, np.concatenate(Es))
    e_offs = [0, len(Es[0]), len(Es[0]) + len(Es[1])]
    dotp = None
    if with_dotp:
        N = n
        derefs = fast_table(rng, 6 * N, zero_frac=0.05)
        vals = fast_table(rng, 3 * N, zero_frac=0.4)
        dotp = [dev_table(ctx, derefs), dev_table(ctx, vals), dev_table(ctx, rows=18 * (N // 4))]
        hN = N // 2
        D = []  # per dot-product circuit half k = 2m + half: (L, R, W)
        for kk in range(6):
            m, half = kk >> 1, kk & 1
            D.append([derefs[m * N + half * hN: m * N + (half + 1) * hN].copy(),
                      derefs[(3 + m) * N + half * hN: (3 + m) * N + (half + 1) * hN].copy(),
                      vals[m * N + half * hN: m * N + (half + 1) * hN].copy()])
    length = h
    r = None
    for j in range(3):
        if j:
            A = [O.bound_top(a, r) for a in A]
            B = [O.bound_top(b, r) for b in B]
            if with_dotp:
                D = [[O.bound_top(x, r) for x in trip] for trip in D]
        got = batched_round(ctx, forest_v, n, ncirc, length, E_v, e_offs[j], r, lead, dotp, first_fold=int(j == 1))
        z = zeros[:len(A[0])]
        for t in range(ncirc):
            exp = O.sc_cubic_round(dup(Es[j]), A[t], B[t], np.ascontiguousarray(z))
            if lead:
                t0, c2, _ = lead_from_evals(*[to_int(x) for x in exp])
                assert [to_int(got[t][0]), to_int(got[t][1])] == [t0, c2], f"round {j} circuit {t}"
            else:
                assert np.array_equal(got[t], exp), f"round {j} circuit {t}"
        if with_dotp:
            for kk in range(6):
                exp = O.sc_cubic_round(D[kk][0], D[kk][1], D[kk][2], np.ascontiguousarray(z))
                assert np.array_equal(got[12 + kk], exp), f"round {j} dot-product circuit {kk}"
        if j:
            length //= 2
        r = rand_scalar(rng)
    # folded tables on the device = the oracle's folds (live length after two binds: h/4)
    live = len(A[0])
    for t in range(ncirc):
        o = t * 2 * n
        assert np.array_equal(forest_v.read(o, live), A[t]) and np.array_equal(forest_v.read(o + h, live), B[t]), f"circuit {t}"
    if with_dotp:
        q4 = N // 4
        for kk in range(6):
            for tt in range(3):
                assert np.array_equal(dotp[2].read((3 * kk + tt) * q4, live), D[kk][tt])
        # the committed polynomials were not touched by the first fold
        assert np.array_equal(dotp[0].read(0, 6 * N), derefs)
    for t in [forest_v, E_v] + (dotp or []):
        t.free()


# ---- DensePolynomial::bound and the hash layer's one-pass slice evaluation (poly.hip) ------------------------------------

def _oracle_poly_bound(Z, Lv, Ls, Rs):
    out = np.zeros((Rs, 4), dtype=np.uint64)
    O.lib().oracle_poly_bound(O.ptr(np.ascontiguousarray(Z)), O.ptr(np.ascontiguousarray(Lv)), Ls, Rs, O.ptr(out))
    return out


def _oracle_evaluate(Z, r):
    """DensePolynomial::evaluate through the oracle's bound_poly_var_top, first challenge = top variable"""
    t = np.array(Z, dtype=np.uint64)
    for k in range(len(r)):
        t = O.bound_top(t, r[k])
    return t[0]


@pytest.mark.parametrize("ell_slice,nbits,used,zero_frac", [(16, 3, 6, 0.0), (15, 4, 15, 0.3), (17, 1, 2, 0.0), (6, 3, 6, 0.0), (4, 4, 16, 0.5),
                                                             (20, 3, 6, 0.1)])
def test_slice_pass_vs_oracle(ctx, ell_slice, nbits, used, zero_frac):
    """vpin_poly_slices_bound (slices_bound_kernel / slices_eval_kernel / slices_combine_kernel): the slice evaluations equal the
    oracle's DensePolynomial::evaluate of every slice, and the LZ vector equals the oracle's DensePolynomial::bound
    (dense_mlpoly.rs:220-227) of the whole table at L = eq((ch, r)[..left]) -- what the two-pass path computed; also
    vpin_poly_bound itself against the same oracle function."""
    rng = np.random.default_rng(1000 * ell_slice + nbits)
    N, S = 1 << ell_slice, 1 << nbits
    Z = fast_table(rng, N * S, zero_frac)
    Z[used * N:] = 0                      # the reference pads the combined polynomials with zero slices
    r = np.stack([rand_scalar(rng) for _ in range(ell_slice)])
    ch = np.stack([rand_scalar(rng) for _ in range(nbits)])
    t = ctx.upload(Z)
    ev, lz = ctx.poly_slices_bound(t, nbits, used, r, ch)
    for s in range(used):
        assert np.array_equal(ev[s], _oracle_evaluate(Z[s * N:(s + 1) * N], r)), f"slice {s}"
    ell = ell_slice + nbits
    left = ell // 2
    Ls, Rs = 1 << left, 1 << (ell - left)
    Lv = O.eq_evals(np.concatenate([ch, r])[:left])
    exp = _oracle_poly_bound(Z, Lv, Ls, Rs)
    assert np.array_equal(lz, exp)
    assert np.array_equal(ctx.poly_bound(t, Lv), exp)
    t.free()


@pytest.mark.parametrize("ell_slice,nbits,n32,used", [(14, 4, 12, 15), (16, 1, 2, 2), (5, 4, 12, 15), (18, 4, 12, 15)])
def test_slice_pass_u32_vs_field_images(ctx, ell_slice, nbits, n32, used):
    """slices_bound_u32_kernel: addresses / timestamps as u32 (0, 1, 2^32 - 1, runs of zeros among them) give the elements the
    field-image pass gives over Scalar::from(v) -- itself checked against the oracle above -- and the oracle's evaluate."""
    rng = np.random.default_rng(77 + ell_slice)
    N, S = 1 << ell_slice, 1 << nbits
    u = rng.integers(0, 2**32, size=(n32, N), dtype=np.uint64).astype(np.uint32)
    u[0, : N // 4] = 0
    u[1, :] = rng.integers(0, 4, size=N, dtype=np.uint64).astype(np.uint32)     # small timestamps
    u[n32 - 1, ::3] = 0xFFFFFFFF
    fqs = fast_table(rng, (used - n32) * N, 0.2) if used > n32 else None
    # the field images: v * R mod q as Montgomery limbs
    images = M.ints_to_table([int(v) for v in u.reshape(-1)]) if N <= (1 << 14) else None
    r = np.stack([rand_scalar(rng) for _ in range(ell_slice)])
    ch = np.stack([rand_scalar(rng) for _ in range(nbits)])
    tf = ctx.upload(np.concatenate([fqs, np.zeros(((1 << int(np.ceil(np.log2(max(1, used - n32))))) * N - len(fqs), 4), dtype=np.uint64)])) if fqs is not None else None
    ev, lz = ctx.poly_slices_bound_u32(u, tf, nbits, used, r, ch)
    if images is not None:
        Z = np.zeros((S * N, 4), dtype=np.uint64)
        Z[: n32 * N] = images
        if fqs is not None:
            Z[n32 * N: used * N] = fqs
        t = ctx.upload(Z)
        ev2, lz2 = ctx.poly_slices_bound(t, nbits, used, r, ch)
        assert np.array_equal(ev, ev2) and np.array_equal(lz, lz2)
        for s in (0, 1, n32 - 1):
            assert np.array_equal(ev[s], _oracle_evaluate(Z[s * N:(s + 1) * N], r)), f"slice {s}"
        t.free()
    else:
        # too long for the Python image builder: the first and the last u32 slice against the oracle through a device conversion-free
        # identity -- a slice of constant value v evaluates to v (the eq weights sum to one)
        const = np.full((n32, N), 0xFFFFFFFF, dtype=np.uint32)
        const[1, :] = 7
        evc, _ = ctx.poly_slices_bound_u32(const, tf, nbits, used, r, ch)
        assert np.array_equal(evc[0], M.ints_to_table([0xFFFFFFFF])[0]) and np.array_equal(evc[1], M.ints_to_table([7])[0])
    if tf is not None:
        tf.free()
