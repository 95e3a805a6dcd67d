"""Pin the C oracle's F_q against (a) the reference's own known-answer tests
(Spartan/src/scalar/ristretto255.rs:789-1213, transcribed as data in
tests/golden/fq_kat.json) and (b) an independent Python big-integer model."""
import ctypes as C
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


def limbs(hexes):
    return [int(h, 16) for h in hexes]


def fq(lst):
    return O.fq_from_limbs(lst)


def canon(f):
    return M.from_mont_limbs(f.limbs())


def mont(x):
    return fq(M.to_mont_limbs(x))


def raw4(lst):
    return (C.c_uint64 * 4)(*lst)


def u8(lst):
    return (C.c_uint8 * len(lst))(*lst)


def test_constants(kat):
    L = O.lib()
    assert limbs(kat["MODULUS_limbs"]) == [(Q >> (64 * i)) & (2**64 - 1) for i in range(4)]
    assert limbs(kat["R_limbs"]) == M.to_mont_limbs(1)
    assert limbs(kat["R2_limbs"]) == M.to_mont_limbs(2**256 % Q)
    assert limbs(kat["R3_limbs"]) == M.to_mont_limbs(2**512 % Q)
    # test_inv (ristretto255.rs:789-801)
    inv = 1
    for _ in range(63):
        inv = (inv * inv) % 2**64
        inv = (inv * limbs(kat["MODULUS_limbs"])[0]) % 2**64
    assert (-inv) % 2**64 == int(kat["INV"], 16)
    for name in ("FQ_MODULUS", "FQ_R", "FQ_R2", "FQ_R3"):
        got = O.Fq.in_dll(L, name).limbs()
        assert got == limbs(kat[name[3:] + "_limbs"])


def test_to_from_bytes(kat):
    L = O.lib()
    one, zero = fq(limbs(kat["R_limbs"])), fq([0, 0, 0, 0])
    r2 = fq(limbs(kat["R2_limbs"]))
    m1 = L.fq_neg(one)
    for f, key in ((zero, "zero"), (one, "one"), (r2, "R2"), (m1, "minus_one")):
        out = (C.c_uint8 * 32)()
        L.fq_to_bytes(out, f)
        assert list(out) == kat["to_bytes"][key]
        back = O.Fq()
        assert L.fq_from_bytes(back, u8(kat["to_bytes"][key])) == 1
        assert back.limbs() == f.limbs()
    for bad in kat["from_bytes_invalid"]:
        back = O.Fq()
        assert L.fq_from_bytes(back, u8(bad)) == 0
    # the modulus itself is rejected, q-1 accepted
    back = O.Fq()
    assert L.fq_from_bytes(back, u8(list(Q.to_bytes(32, "little")))) == 0
    assert L.fq_from_bytes(back, u8(list((Q - 1).to_bytes(32, "little")))) == 1


def test_from_u512(kat):
    L = O.lib()
    mod = limbs(kat["MODULUS_limbs"])
    def wide(l8):
        b = b"".join(int(x).to_bytes(8, "little") for x in l8)
        return L.fq_from_bytes_wide(u8(list(b)))
    assert wide(mod + [0, 0, 0, 0]).limbs() == [0, 0, 0, 0]
    assert wide([1, 0, 0, 0, 0, 0, 0, 0]).limbs() == limbs(kat["R_limbs"])
    assert wide([0, 0, 0, 0, 1, 0, 0, 0]).limbs() == limbs(kat["R2_limbs"])
    mx = 2**64 - 1
    r3, r = fq(limbs(kat["R3_limbs"])), fq(limbs(kat["R_limbs"]))
    assert wide([mx] * 8).limbs() == L.fq_sub(r3, r).limbs()
    # test_from_bytes_wide_r2 / negative_one / maximum
    assert L.fq_from_bytes_wide(u8(kat["to_bytes"]["R2"] + [0] * 32)).limbs() == limbs(kat["R2_limbs"])
    assert L.fq_from_bytes_wide(u8(kat["to_bytes"]["minus_one"] + [0] * 32)).limbs() == L.fq_neg(r).limbs()
    assert (L.fq_from_bytes_wide(u8([0xFF] * 64)).limbs()
            == L.fq_from_raw(raw4(limbs(kat["from_bytes_wide_max_raw"]))).limbs())
    assert canon(L.fq_from_bytes_wide(u8([0xFF] * 64))) == (2**512 - 1) % Q


def test_add_sub_neg_largest(kat):
    L = O.lib()
    largest = fq(limbs(kat["LARGEST_limbs"]))
    assert L.fq_add(largest, largest).limbs() == limbs(kat["LARGEST_plus_LARGEST_limbs"])
    assert L.fq_add(largest, fq([1, 0, 0, 0])).limbs() == [0, 0, 0, 0]
    assert L.fq_neg(largest).limbs() == [1, 0, 0, 0]
    assert L.fq_neg(fq([0, 0, 0, 0])).limbs() == [0, 0, 0, 0]
    assert L.fq_neg(fq([1, 0, 0, 0])).limbs() == limbs(kat["LARGEST_limbs"])
    assert L.fq_sub(largest, largest).limbs() == [0, 0, 0, 0]
    mod = fq(limbs(kat["MODULUS_limbs"]))
    assert L.fq_sub(fq([0, 0, 0, 0]), largest).limbs() == L.fq_sub(mod, largest).limbs()
    z = fq([0, 0, 0, 0])
    assert L.fq_mul(z, z).limbs() == [0, 0, 0, 0]


def test_mul_square_vs_double_and_add(kat):
    # test_multiplication / test_squaring (ristretto255.rs:1084-1139)
    L = O.lib()
    largest = fq(limbs(kat["LARGEST_limbs"]))
    cur = largest
    for _ in range(100):
        tmp = L.fq_mul(cur, cur)
        assert L.fq_square(cur).limbs() == tmp.limbs()
        out = (C.c_uint8 * 32)()
        L.fq_to_bytes(out, cur)
        tmp2 = fq([0, 0, 0, 0])
        for byte in reversed(list(out)):
            for i in range(7, -1, -1):
                tmp2 = L.fq_add(tmp2, tmp2)
                if (byte >> i) & 1:
                    tmp2 = L.fq_add(tmp2, cur)
        assert tmp.limbs() == tmp2.limbs()
        cur = L.fq_add(cur, largest)


def test_inversion(kat):
    L = O.lib()
    one = fq(limbs(kat["R_limbs"]))
    assert L.fq_invert(fq([0, 0, 0, 0])).limbs() == [0, 0, 0, 0]
    assert L.fq_invert(one).limbs() == one.limbs()
    m1 = L.fq_neg(one)
    assert L.fq_invert(m1).limbs() == m1.limbs()
    r2 = fq(limbs(kat["R2_limbs"]))
    tmp = r2
    for _ in range(100):
        assert L.fq_mul(L.fq_invert(tmp), tmp).limbs() == one.limbs()
        tmp = L.fq_add(tmp, r2)
    qm2 = raw4(limbs(kat["q_minus_2"]))
    r1 = one
    for _ in range(20):
        r1n = L.fq_invert(r1)
        assert r1n.limbs() == L.fq_pow_vartime(r1, qm2).limbs()
        assert canon(r1n) == pow(canon(r1), -1, Q)
        r1 = L.fq_add(r1n, one)


def test_from_raw_and_double(kat):
    L = O.lib()
    assert (L.fq_from_raw(raw4(limbs(kat["from_raw_all_ones_equiv_raw"]))).limbs()
            == L.fq_from_raw(raw4([2**64 - 1] * 4)).limbs())
    assert L.fq_from_raw(raw4(limbs(kat["MODULUS_limbs"]))).limbs() == [0, 0, 0, 0]
    assert L.fq_from_raw(raw4([1, 0, 0, 0])).limbs() == limbs(kat["R_limbs"])
    a = L.fq_from_raw(raw4(limbs(kat["double_input_raw"])))
    av = sum(v << (64 * i) for i, v in enumerate(limbs(kat["double_input_raw"]))) % Q
    assert canon(a) == av
    assert canon(L.fq_add(a, a)) == 2 * av % Q


def test_random_vs_bigint():
    L = O.lib()
    rng = random.Random(1234)
    for _ in range(300):
        x, y = rng.randrange(Q), rng.randrange(Q)
        fx, fy = mont(x), mont(y)
        assert canon(L.fq_add(fx, fy)) == (x + y) % Q
        assert canon(L.fq_sub(fx, fy)) == (x - y) % Q
        assert canon(L.fq_mul(fx, fy)) == (x * y) % Q
        assert canon(L.fq_neg(fx)) == (-x) % Q
        w = rng.randrange(2**512)
        assert canon(L.fq_from_bytes_wide(u8(list(w.to_bytes(64, "little"))))) == w % Q
        v = rng.randrange(2**256)
        assert canon(L.fq_from_bytes_mod_order(u8(list(v.to_bytes(32, "little"))))) == v % Q
    xs = [rng.randrange(1, Q) for _ in range(17)]
    arr = M.ints_to_table(xs)
    ret = L.fq_batch_invert(O.ptr(arr), len(xs))
    assert M.table_to_ints(arr) == [pow(x, -1, Q) for x in xs]
    prod = 1
    for x in xs:
        prod = prod * x % Q
    assert canon(ret) == pow(prod, -1, Q)
