"""Independent Python model of vPIN's two R1CS gadgets and of libspartan's Instance::new
padding, used by the tests as (a) the instance generator for oracle runs and (b) a
cross-check of the product's C++ gadget builder.

Follows vPIN_proof_generation/src/point_addition.rs:67-327, point_mult.rs:61-704 and
Spartan/src/lib.rs:138-244.  Curve E2: src/convolution/Client.py:134-143.
"""
import numpy as np

import pymodel as M

Q = M.Q

# E2: y^2 = x^3 + a x + b over F_q (src/convolution/Client.py:138-143)
E2_A = 3491403595575449084947959021303599933011749826127899762162894550148391771037
E2_B = 3633908682298454119909199192149978293706667958442512986315258451820769071958
E2_GX = 4561981307020378385254256586024830594940985765081274686120783167106442831732
E2_GY = 684120277165286233470758410892647831027470652988879249692043589061244861334
E2_ORDER = 7237005577332262213973186563042994240704759454384003648147593987722918659549

# a_pd_byte of point_mult.rs:341 (must equal E2_A)
A_PD_BYTES = [157, 27, 50, 101, 63, 42, 38, 142, 68, 159, 245, 15, 16, 47, 75, 58, 203, 87, 15, 3, 219, 183, 77,
              94, 64, 118, 147, 233, 124, 16, 184, 7]
assert int.from_bytes(bytes(A_PD_BYTES), "little") == E2_A


def inv(x):
    """dalek Scalar::invert: x^(q-2); 0 -> 0"""
    return pow(x % Q, Q - 2, Q)


def e2_add(P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    (x1, y1), (x2, y2) = P1, P2
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = (3 * x1 * x1 + E2_A) * inv(2 * y1) % Q
    else:
        lam = (y2 - y1) * inv(x2 - x1) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return x3, (lam * (x1 - x3) - y1) % Q


def e2_mul(k, P):
    acc = None
    while k:
        if k & 1:
            acc = e2_add(acc, P)
        P = e2_add(P, P)
        k >>= 1
    return acc


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    return state, z ^ (z >> 31)


def synthetic_points(seed, count):
    """count points k*G on E2 with k from SplitMix64."""
    G = (E2_GX, E2_GY)
    out, st = [], seed
    for _ in range(count):
        st, k = splitmix64(st)
        out.append(e2_mul(k % E2_ORDER or 1, G))
    return out


# ------------------------------------------------------------------------------------------------

def build_point_add(ops):
    """ops: list of (px, py, rx, ry, rz) ints.  point_addition.rs:67-327."""
    N = len(ops)
    num_cons, num_vars, num_inputs = 10 * N, 15 * N + 1, 0
    A, B, C = [], [], []
    one, m1 = 1, Q - 1
    nv = num_vars
    for i in range(N):
        r, v = 10 * i, 15 * i
        A.append((r + 0, v + 0, one)); B.append((r + 0, v + 1, one)); B.append((r + 0, v + 2, m1)); C.append((r + 0, nv, one))
        A.append((r + 1, v + 3, one)); A.append((r + 1, v + 4, m1)); B.append((r + 1, v + 0, one)); C.append((r + 1, v + 6, one))
        A.append((r + 2, v + 6, one)); B.append((r + 2, v + 6, one)); C.append((r + 2, v + 7, one))
        A.append((r + 3, v + 7, one)); A.append((r + 3, v + 2, m1)); A.append((r + 3, v + 1, m1))
        B.append((r + 3, nv, one)); B.append((r + 3, v + 5, m1)); C.append((r + 3, v + 9, one))
        A.append((r + 4, v + 2, one)); B.append((r + 4, v + 5, one)); C.append((r + 4, v + 10, one))
        A.append((r + 5, v + 9, one)); A.append((r + 5, v + 10, one)); B.append((r + 5, nv, one)); C.append((r + 5, v + 13, one))
        A.append((r + 6, v + 6, one)); B.append((r + 6, v + 2, one)); B.append((r + 6, v + 13, m1)); C.append((r + 6, v + 8, one))
        A.append((r + 7, v + 8, one)); A.append((r + 7, v + 4, m1)); B.append((r + 7, nv, one)); B.append((r + 7, v + 5, m1))
        C.append((r + 7, v + 11, one))
        A.append((r + 8, v + 4, one)); B.append((r + 8, v + 5, one)); C.append((r + 8, v + 12, one))
        A.append((r + 9, v + 11, one)); A.append((r + 9, v + 12, one)); B.append((r + 9, nv, one)); C.append((r + 9, v + 14, one))
    vars_input = [0] * num_vars
    for i, (px, py, rx, ry, rz) in enumerate(ops):
        c = inv(rx - px)
        s1 = (ry - py) * c % Q
        s2 = s1 * s1 % Q
        t1 = (s2 - px - rx) * (1 - rz) % Q
        t2 = px * rz % Q
        x3 = (t1 + t2) % Q
        s3 = s1 * (px - x3) % Q
        t3 = (s3 - py) * (1 - rz) % Q
        t4 = py * rz % Q
        y3 = (t3 + t4) % Q
        vars_input[15 * i:15 * i + 15] = [c, rx % Q, px % Q, ry % Q, py % Q, rz % Q, s1, s2, s3, t1, t2, t3, t4, x3, y3]
    vars_para = [0] * num_vars
    return dict(num_cons=num_cons, num_vars=num_vars, num_inputs=num_inputs, A=A, B=B, C=C,
                vars_para=vars_para, vars_input=vars_input, vars=list(vars_input), inputs=[])


def _pa(bx, by, bz, ax, ay):
    c = inv(bx - ax)
    s1 = (by - ay) * c % Q
    s2 = s1 * s1 % Q
    t1 = (s2 - ax - bx) * (1 - bz) % Q
    t2 = ax * bz % Q
    cx = (t1 + t2) % Q
    s3 = s1 * (ax - cx) % Q
    t3 = (s3 - ay) * (1 - bz) % Q
    t4 = ay * bz % Q
    cy = (t3 + t4) % Q
    return cx, cy, c, s1, s2, s3, t1, t2, t3, t4


def _pd(ax, ay, a):
    c = inv(2 * ay)
    t1 = ax * ax % Q
    s1 = (3 * t1 + a) * c % Q
    s2 = s1 * s1 % Q
    dx = (s2 - 2 * ax) % Q
    t2 = s1 * (ax - dx) % Q
    dy = (t2 - ay) % Q
    return dx, dy, t1, t2, s1, s2, c


def build_point_mult(ops, n=128):
    """ops: list of (weight:int < 2^128, px, py).  point_mult.rs:61-704 (n = 128, load_data.rs:62)."""
    N = len(ops)
    oc, ov = 27 * n + 8, n + 10 + n * 26
    num_cons, num_vars, num_inputs = oc * N, ov * N + 1, 1
    nv = num_vars
    A, B, C = [], [], []
    one, two, three, m1, m2 = 1, 2, 3, Q - 1, Q - 2
    for j in range(N):
        r0, v0 = oc * j, ov * j
        tb = 1
        for i in range(n):
            A.append((r0, v0 + i, tb))
            tb = tb * 2 % Q
        B.append((r0, nv, one)); C.append((r0, v0 + n, one))
        for i in range(1, n + 1):
            A.append((r0 + i, v0 + i - 1, one)); B.append((r0 + i, v0 + i - 1, one)); C.append((r0 + i, v0 + i - 1, one))
        A.append((r0 + n + 1, v0 + n + 1, one)); A.append((r0 + n + 1, v0 + 10 * n + 8, m1)); B.append((r0 + n + 1, nv, one))
        A.append((r0 + n + 2, v0 + 2 * n + 2, one)); A.append((r0 + n + 2, v0 + 10 * n + 9, m1)); B.append((r0 + n + 2, nv, one))
        A.append((r0 + n + 3, v0 + 3 * n + 3, one)); B.append((r0 + n + 3, nv, one))
        A.append((r0 + n + 4, v0 + 4 * n + 4, one)); B.append((r0 + n + 4, nv, one))
        A.append((r0 + n + 5, v0 + 5 * n + 5, one)); A.append((r0 + n + 5, nv, m1)); B.append((r0 + n + 5, nv, one))
        for i in range(n):
            r = r0 + i * 26
            v = v0 + i
            # PA
            A.append((r + n + 6, v + 10 * n + 10, one)); B.append((r + n + 6, v + 3 * n + 3, one)); B.append((r + n + 6, v + n + 1, m1)); C.append((r + n + 6, nv, one))
            A.append((r + n + 7, v + 4 * n + 4, one)); A.append((r + n + 7, v + 2 * n + 2, m1)); B.append((r + n + 7, v + 10 * n + 10, one)); C.append((r + n + 7, v + 11 * n + 10, one))
            A.append((r + n + 8, v + 11 * n + 10, one)); B.append((r + n + 8, v + 11 * n + 10, one)); C.append((r + n + 8, v + 12 * n + 10, one))
            A.append((r + n + 9, v + 12 * n + 10, one)); A.append((r + n + 9, v + n + 1, m1)); A.append((r + n + 9, v + 3 * n + 3, m1))
            B.append((r + n + 9, nv, one)); B.append((r + n + 9, v + 5 * n + 5, m1)); C.append((r + n + 9, v + 14 * n + 10, one))
            A.append((r + n + 10, v + n + 1, one)); B.append((r + n + 10, v + 5 * n + 5, one)); C.append((r + n + 10, v + 15 * n + 10, one))
            A.append((r + n + 11, v + 14 * n + 10, one)); A.append((r + n + 11, v + 15 * n + 10, one)); B.append((r + n + 11, nv, one)); C.append((r + n + 11, v + 6 * n + 6, one))
            A.append((r + n + 12, v + 11 * n + 10, one)); B.append((r + n + 12, v + n + 1, one)); B.append((r + n + 12, v + 6 * n + 6, m1)); C.append((r + n + 12, v + 13 * n + 10, one))
            A.append((r + n + 13, v + 13 * n + 10, one)); A.append((r + n + 13, v + 2 * n + 2, m1)); B.append((r + n + 13, nv, one)); B.append((r + n + 13, v + 5 * n + 5, m1)); C.append((r + n + 13, v + 16 * n + 10, one))
            A.append((r + n + 14, v + 2 * n + 2, one)); B.append((r + n + 14, v + 5 * n + 5, one)); C.append((r + n + 14, v + 17 * n + 10, one))
            A.append((r + n + 15, v + 16 * n + 10, one)); A.append((r + n + 15, v + 17 * n + 10, one)); B.append((r + n + 15, nv, one)); C.append((r + n + 15, v + 7 * n + 6, one))
            # PD
            A.append((r + n + 16, v + 18 * n + 10, one)); B.append((r + n + 16, v + 2 * n + 2, two)); C.append((r + n + 16, nv, one))
            A.append((r + n + 17, v + n + 1, one)); B.append((r + n + 17, v + n + 1, one)); C.append((r + n + 17, v + 19 * n + 10, one))
            A.append((r + n + 18, v + 19 * n + 10, three)); A.append((r + n + 18, nv + 1, one)); B.append((r + n + 18, v + 18 * n + 10, one)); C.append((r + n + 18, v + 20 * n + 10, one))
            A.append((r + n + 19, v + 20 * n + 10, one)); B.append((r + n + 19, v + 20 * n + 10, one)); C.append((r + n + 19, v + 21 * n + 10, one))
            A.append((r + n + 20, v + 21 * n + 10, one)); A.append((r + n + 20, v + n + 1, m2)); B.append((r + n + 20, nv, one)); C.append((r + n + 20, v + 8 * n + 6, one))
            A.append((r + n + 21, v + 20 * n + 10, one)); B.append((r + n + 21, v + n + 1, one)); B.append((r + n + 21, v + 8 * n + 6, m1)); C.append((r + n + 21, v + 22 * n + 10, one))
            A.append((r + n + 22, v + 22 * n + 10, one)); A.append((r + n + 22, v + 2 * n + 2, m1)); B.append((r + n + 22, nv, one)); C.append((r + n + 22, v + 9 * n + 6, one))
            # select
            A.append((r + n + 23, v + 6 * n + 6, one)); B.append((r + n + 23, v, one)); C.append((r + n + 23, v + 23 * n + 10, one))
            A.append((r + n + 24, v + 3 * n + 3, one)); B.append((r + n + 24, nv, one)); B.append((r + n + 24, v, m1)); C.append((r + n + 24, v + 24 * n + 10, one))
            A.append((r + n + 25, v + 23 * n + 10, one)); A.append((r + n + 25, v + 24 * n + 10, one)); B.append((r + n + 25, nv, one)); C.append((r + n + 25, v + 3 * n + 4, one))
            A.append((r + n + 26, v + 7 * n + 6, one)); B.append((r + n + 26, v, one)); C.append((r + n + 26, v + 25 * n + 10, one))
            A.append((r + n + 27, v + 4 * n + 4, one)); B.append((r + n + 27, nv, one)); B.append((r + n + 27, v, m1)); C.append((r + n + 27, v + 26 * n + 10, one))
            A.append((r + n + 28, v + 25 * n + 10, one)); A.append((r + n + 28, v + 26 * n + 10, one)); B.append((r + n + 28, nv, one)); C.append((r + n + 28, v + 4 * n + 5, one))
            A.append((r + n + 29, v + 5 * n + 5, one)); B.append((r + n + 29, nv, one)); B.append((r + n + 29, v, m1)); C.append((r + n + 29, v + 5 * n + 6, one))
            A.append((r + n + 30, v + n + 2, one)); A.append((r + n + 30, v + 8 * n + 6, m1)); B.append((r + n + 30, nv, one))
            A.append((r + n + 31, v + 2 * n + 3, one)); A.append((r + n + 31, v + 9 * n + 6, m1)); B.append((r + n + 31, nv, one))
        A.append((r0 + oc - 2, v0 + 10 * n + 6, one)); A.append((r0 + oc - 2, v0 + 3 * n + 3 + n, m1)); B.append((r0 + oc - 2, nv, one))
        A.append((r0 + oc - 1, v0 + 10 * n + 7, one)); A.append((r0 + oc - 1, v0 + 4 * n + 4 + n, m1)); B.append((r0 + oc - 1, nv, one))

    vars_para, vars_input = [0] * num_vars, [0] * num_vars
    for j, (w, px, py) in enumerate(ops):
        v0 = ov * j
        w &= (1 << 128) - 1
        bits = [(w >> i) & 1 for i in range(n)]
        ax_prev, ay_prev = px % Q, py % Q
        bx_prev, by_prev, bz_prev = 0, 0, 1
        vars_para[v0 + n] = w % Q
        vi = vars_input
        vi[v0 + n + 1] = ax_prev; vi[v0 + 2 * n + 2] = ay_prev
        vi[v0 + 3 * n + 3] = 0; vi[v0 + 4 * n + 4] = 0; vi[v0 + 5 * n + 5] = 1
        vi[v0 + 10 * n + 8] = px % Q; vi[v0 + 10 * n + 9] = py % Q
        for i in range(n):
            cx, cy, c_pa, s1_pa, s2_pa, s3_pa, t1_pa, t2_pa, t3_pa, t4_pa = _pa(bx_prev, by_prev, bz_prev, ax_prev, ay_prev)
            dx, dy, t1_pd, t2_pd, s1_pd, s2_pd, c_pd = _pd(ax_prev, ay_prev, E2_A)
            b = bits[i]
            z1 = cx * b % Q; z2 = bx_prev * (1 - b) % Q; bx = (z1 + z2) % Q
            z3 = cy * b % Q; z4 = by_prev * (1 - b) % Q; by = (z3 + z4) % Q
            bz = bz_prev * (1 - b) % Q
            vi[v0 + i] = b
            vi[v0 + n + 2 + i] = dx; vi[v0 + 2 * n + 3 + i] = dy
            vi[v0 + 3 * n + 4 + i] = bx; vi[v0 + 4 * n + 5 + i] = by; vi[v0 + 5 * n + 6 + i] = bz
            vi[v0 + 6 * n + 6 + i] = cx; vi[v0 + 7 * n + 6 + i] = cy
            vi[v0 + 8 * n + 6 + i] = dx; vi[v0 + 9 * n + 6 + i] = dy
            vi[v0 + 10 * n + 10 + i] = c_pa; vi[v0 + 11 * n + 10 + i] = s1_pa; vi[v0 + 12 * n + 10 + i] = s2_pa
            vi[v0 + 13 * n + 10 + i] = s3_pa; vi[v0 + 14 * n + 10 + i] = t1_pa; vi[v0 + 15 * n + 10 + i] = t2_pa
            vi[v0 + 16 * n + 10 + i] = t3_pa; vi[v0 + 17 * n + 10 + i] = t4_pa
            vi[v0 + 18 * n + 10 + i] = c_pd; vi[v0 + 19 * n + 10 + i] = t1_pd; vi[v0 + 20 * n + 10 + i] = s1_pd
            vi[v0 + 21 * n + 10 + i] = s2_pd; vi[v0 + 22 * n + 10 + i] = t2_pd
            vi[v0 + 23 * n + 10 + i] = z1; vi[v0 + 24 * n + 10 + i] = z2
            vi[v0 + 25 * n + 10 + i] = z3; vi[v0 + 26 * n + 10 + i] = z4
            ax_prev, ay_prev, bx_prev, by_prev, bz_prev = dx, dy, bx, by, bz
        vi[v0 + 10 * n + 6] = bx_prev; vi[v0 + 10 * n + 7] = by_prev
    vars_all = [(a + b) % Q for a, b in zip(vars_para, vars_input)]
    return dict(num_cons=num_cons, num_vars=num_vars, num_inputs=num_inputs, A=A, B=B, C=C,
                vars_para=vars_para, vars_input=vars_input, vars=vars_all, inputs=[E2_A])


# ------------------------------------------------------------------------------------------------

def next_pow2(x):
    p = 1
    while p < x:
        p *= 2
    return p


def instance_new(g):
    """Instance::new padding + column remap (lib.rs:138-244) and assignment padding.
    Returns a dict of numpy arrays ready for the oracle / the C ABI."""
    nc, nvr, ni = g["num_cons"], g["num_vars"], g["num_inputs"]
    nv_pad = next_pow2(max(nvr, ni + 1))
    nc_pad = 2 if nc in (0, 1) else next_pow2(nc)
    out = dict(num_cons=nc_pad, num_vars=nv_pad, num_inputs=ni, num_cons_unpadded=nc, num_vars_unpadded=nvr)
    for name in "ABC":
        trip = g[name]
        rows = np.array([t[0] for t in trip], dtype=np.uint32)
        cols = np.array([t[1] + (nv_pad - nvr if t[1] >= nvr else 0) for t in trip], dtype=np.uint32)
        # few distinct values: convert through a cache
        cache = {}
        vals = np.zeros((len(trip), 4), dtype=np.uint64)
        for k, t in enumerate(trip):
            v = t[2] % Q
            if v not in cache:
                cache[v] = M.to_mont_limbs(v)
            vals[k] = cache[v]
        assert rows.max(initial=0) < nc and cols.max(initial=0) < 2 * nv_pad
        out[name] = (rows, cols, vals)
    for key in ("vars_para", "vars_input", "vars"):
        out[key] = M.ints_to_table(g[key] + [0] * (nv_pad - nvr))
    out["inputs"] = M.ints_to_table(g["inputs"]) if g["inputs"] else np.zeros((0, 4), dtype=np.uint64)
    return out


def synthetic_add_ops(seed, count, rz_one_every=0):
    pts = synthetic_points(seed, 2 * count)
    ops = []
    for i in range(count):
        (px, py), (rx, ry) = pts[2 * i], pts[2 * i + 1]
        if rz_one_every and i % rz_one_every == 0:
            ops.append((px, py, 0, 0, 1))  # R = infinity
        else:
            ops.append((px, py, rx, ry, 0))
    return ops


def synthetic_mult_ops(seed, count, weights=None):
    pts = synthetic_points(seed, count)
    st = seed ^ 0xABCDEF
    ops = []
    for i in range(count):
        if weights is not None:
            w = weights[i % len(weights)]
        else:
            st, a = splitmix64(st)
            st, b = splitmix64(st)
            w = ((a << 64) | b) >> 1
        ops.append((w, pts[i][0], pts[i][1]))
    return ops
