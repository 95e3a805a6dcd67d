"""Independent Python big-integer model of the F_q table operations.

Used only by tests (and by tests/golden/make_golden.py to emit fixtures): it is a
second, structurally different statement of the reference formulas
(Spartan/src/dense_mlpoly.rs:78-94,229-236; Spartan/src/sumcheck.rs:460-469,624-652;
Spartan/src/unipoly.rs:23-54) in plain `% q` arithmetic, against which the C oracle
under oracle/ is pinned.
"""
import numpy as np

Q = 2**252 + 27742317777372353535851937790883648493
R = 2**256 % Q
RINV = pow(R, -1, Q)


def to_mont_limbs(x):
    """canonical int -> 4 u64 limbs of x*R mod q (the reference's in-memory Scalar)."""
    m = (x * R) % Q
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_mont_limbs(limbs):
    m = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return (m * RINV) % Q


def ints_to_table(xs):
    """list of canonical ints -> (n,4) uint64 array in Montgomery form."""
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = to_mont_limbs(x)
    return out


def table_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [from_mont_limbs(row) for row in arr]


def eq_evals(r):
    """EqPolynomial::evals (dense_mlpoly.rs:78-94) stated as the tensor product:
    evals[b] = prod_j (r_j if bit_j(b) else 1-r_j), bit 0 of the *index string* is r[0]
    (most significant index bit)."""
    ell = len(r)
    out = []
    for b in range(1 << ell):
        v = 1
        for j in range(ell):
            bit = (b >> (ell - 1 - j)) & 1
            v = v * (r[j] if bit else (1 - r[j])) % Q
        out.append(v)
    return out


def bound_top(Z, r):
    n = len(Z) // 2
    return [(Z[i] + r * (Z[i + n] - Z[i])) % Q for i in range(n)]


def sc_cubic_round(A, B, C, D):
    n = len(A) // 2
    e0 = e2 = e3 = 0
    for i in range(n):
        e0 += A[i] * (B[i] * C[i] - D[i])
        a2, b2, c2, d2 = (2 * X[n + i] - X[i] for X in (A, B, C, D))
        e2 += a2 * (b2 * c2 - d2)
        a3, b3, c3, d3 = (3 * X[n + i] - 2 * X[i] for X in (A, B, C, D))
        e3 += a3 * (b3 * c3 - d3)
    return e0 % Q, e2 % Q, e3 % Q


def sc_quad_round(A, B):
    n = len(A) // 2
    e0 = e2 = 0
    for i in range(n):
        e0 += A[i] * B[i]
        e2 += (2 * A[n + i] - A[i]) * (2 * B[n + i] - B[i])
    return e0 % Q, e2 % Q


def unipoly_from_evals(evals):
    """Lagrange interpolation through x = 0..deg (constant term first)."""
    n = len(evals)
    coeffs = [0] * n
    for i in range(n):
        # basis polynomial l_i(x) = prod_{j!=i} (x - j)/(i - j)
        num = [1]
        den = 1
        for j in range(n):
            if j == i:
                continue
            num = [(a - j * b) % Q for a, b in zip([0] + num, num + [0])]
            den = den * (i - j) % Q
        s = evals[i] * pow(den, -1, Q) % Q
        for k in range(n):
            coeffs[k] = (coeffs[k] + s * num[k]) % Q
    return coeffs


def unipoly_eval(coeffs, r):
    return sum(c * pow(r, k, Q) for k, c in enumerate(coeffs)) % Q


def mle_eval(Z, r):
    chis = eq_evals(r)
    return sum(z * c for z, c in zip(Z, chis)) % Q
