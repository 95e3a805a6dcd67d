"""Pin the C oracle's table ops / sum-check round reductions against the reference's
own vectors (dense_mlpoly.rs:448-466, unipoly.rs:128-181, SURVEY A.6 micro-KAT) and the
independent Python big-integer model (tests/pymodel.py)."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import pymodel as M

Q = M.Q


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "fq_kat.json")) as f:
        return json.load(f)


def T(xs):
    return M.ints_to_table([x % Q for x in xs])


def test_micro_kat(kat):
    k = kat["sumcheck_micro"]
    tau, A, B, Cc = T(k["tau"]), T(k["Az"]), T(k["Bz"]), T(k["Cz"])
    out = O.sc_cubic_round(tau, A, B, Cc)
    assert M.table_to_ints(out) == [k["e0"], k["e2"], k["e3"]]
    r = T([k["r"]])[0]
    for tab, key in ((tau, "tau_f"), (A, "Az_f"), (B, "Bz_f"), (Cc, "Cz_f")):
        assert M.table_to_ints(O.bound_top(tab, r)) == k[key]


def test_mle_and_unipoly_kat(kat):
    L = O.lib()
    m = kat["mle"]
    Z, r = T(m["Z"]), T(m["r"])
    assert M.from_mont_limbs(L.oracle_poly_evaluate(O.ptr(Z), O.ptr(r), 2).limbs()) == m["eval"]
    for key, pt, ev in (("quad", 3, "eval_at_3"), ("cubic", 4, "eval_at_4")):
        u = kat["unipoly"][key]
        evals = T(u["evals"])
        coeffs = np.zeros_like(evals)
        L.oracle_unipoly_from_evals(O.ptr(evals), len(u["evals"]), O.ptr(coeffs))
        assert M.table_to_ints(coeffs) == u["coeffs"]
        rr = T([pt])
        got = L.oracle_unipoly_evaluate(O.ptr(coeffs), len(u["coeffs"]), O.ptr(rr))
        assert M.from_mont_limbs(got.limbs()) == u[ev]


@pytest.mark.parametrize("ell", [1, 2, 5, 8])
def test_random_rounds_vs_model(ell):
    rng = random.Random(100 + ell)
    n = 1 << ell
    tabs_i = [[rng.randrange(Q) for _ in range(n)] for _ in range(4)]
    tabs = [T(t) for t in tabs_i]
    while n >= 2:
        got = M.table_to_ints(O.sc_cubic_round(*[t[:n] for t in tabs]))
        assert got == list(M.sc_cubic_round(*[t[:n] for t in tabs_i]))
        gotq = M.table_to_ints(O.sc_quad_round(tabs[0][:n], tabs[1][:n]))
        assert gotq == list(M.sc_quad_round(tabs_i[0][:n], tabs_i[1][:n]))
        r = rng.randrange(Q)
        tabs = [O.bound_top(t[:n], T([r])[0]) for t in tabs]
        tabs_i = [M.bound_top(t[:n], r) for t in tabs_i]
        for a, b in zip(tabs, tabs_i):
            assert M.table_to_ints(a) == b
        n //= 2


@pytest.mark.parametrize("ell", [0, 1, 3, 6])
def test_eq_evals_and_mle(ell):
    rng = random.Random(7 + ell)
    r = [rng.randrange(Q) for _ in range(ell)]
    got = O.eq_evals(T(r)) if ell else np.array([M.to_mont_limbs(1)], dtype=np.uint64)
    assert M.table_to_ints(got) == M.eq_evals(r)
    Z = [rng.randrange(Q) for _ in range(1 << ell)]
    if ell:
        L = O.lib()
        zt, rt = T(Z), T(r)
        v = L.oracle_poly_evaluate(O.ptr(zt), O.ptr(rt), ell)
        assert M.from_mont_limbs(v.limbs()) == M.mle_eval(Z, r)
        # folding every variable top-down gives the same evaluation (dense_mlpoly.rs:537-579 idea)
        cur = Z
        for rj in r:
            cur = M.bound_top(cur, rj)
        assert cur[0] == M.mle_eval(Z, r)


def test_unipoly_random():
    L = O.lib()
    rng = random.Random(5)
    for n in (3, 4):
        for _ in range(20):
            ev = [rng.randrange(Q) for _ in range(n)]
            evals = T(ev)
            coeffs = np.zeros_like(evals)
            L.oracle_unipoly_from_evals(O.ptr(evals), n, O.ptr(coeffs))
            assert M.table_to_ints(coeffs) == M.unipoly_from_evals(ev)
            c = M.table_to_ints(coeffs)
            for x in range(n):
                assert M.unipoly_eval(c, x) == ev[x]


def test_poly_bound_LZ():
    L = O.lib()
    rng = random.Random(9)
    Ls, Rs = 4, 8
    Z = [rng.randrange(Q) for _ in range(Ls * Rs)]
    Lv = [rng.randrange(Q) for _ in range(Ls)]
    zt, lt = T(Z), T(Lv)
    out = np.zeros((Rs, 4), dtype=np.uint64)
    L.oracle_poly_bound(O.ptr(zt), O.ptr(lt), Ls, Rs, O.ptr(out))
    exp = [sum(Lv[j] * Z[j * Rs + i] for j in range(Ls)) % Q for i in range(Rs)]
    assert M.table_to_ints(out) == exp
