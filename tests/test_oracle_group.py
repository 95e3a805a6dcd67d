"""Pin the oracle's ristretto255 / SHAKE256 / Merlin restatements against public vectors
(RFC 9496 App. A, Merlin conformance vector, hashlib.shake_256) and the Python model."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import pymodel as M
import pymodel_group as G
from oracle_lib import Ge, Merlin, Shake


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "ristretto_kat.json")) as f:
        return json.load(f)


def comp(L, g):
    out = (C.c_uint8 * 32)()
    L.ge_compress(out, C.byref(g))
    return bytes(out)


def u8(b):
    return (C.c_uint8 * len(b))(*b)


def test_generator_multiples(kat):
    L = O.lib()
    B = Ge()
    L.ge_basepoint(C.byref(B))
    acc = Ge()
    L.ge_identity(C.byref(acc))
    pyB, pyacc = G.basepoint(), G.Pt.identity()
    for k in range(0, 17):
        enc = comp(L, acc)
        assert enc == pyacc.encode()
        if str(k) in kat["multiples_of_generator"]:
            assert enc.hex() == kat["multiples_of_generator"][str(k)]
        dec = Ge()
        assert L.ge_decompress(C.byref(dec), u8(enc)) == 1
        assert L.ge_eq(C.byref(dec), C.byref(acc)) == 1
        assert comp(L, dec) == enc
        L.ge_add(C.byref(acc), C.byref(acc), C.byref(B))
        pyacc = pyacc + pyB
    d, a = Ge(), Ge()
    L.ge_double(C.byref(d), C.byref(B))
    L.ge_add(C.byref(a), C.byref(B), C.byref(B))
    assert comp(L, d) == comp(L, a) == bytes.fromhex(kat["multiples_of_generator"]["2"])


def test_bad_encodings_rejected(kat):
    L = O.lib()
    for h in kat["bad_encodings"]:
        g = Ge()
        assert L.ge_decompress(C.byref(g), u8(bytes.fromhex(h))) == 0
        assert G.decode(bytes.fromhex(h)) is None


def test_hash_to_group(kat):
    L = O.lib()
    for label, exp in kat["hash_to_group_sha512"]:
        h = hashlib.sha512(label.encode()).digest()
        g = Ge()
        L.ge_from_uniform_bytes(C.byref(g), u8(h))
        assert comp(L, g).hex() == exp
        assert G.from_uniform_bytes(h).encode().hex() == exp
    rng = random.Random(3)
    for _ in range(20):
        h = bytes(rng.randrange(256) for _ in range(64))
        g = Ge()
        L.ge_from_uniform_bytes(C.byref(g), u8(h))
        assert comp(L, g) == G.from_uniform_bytes(h).encode()


def test_shake256_vs_hashlib():
    L = O.lib()
    rng = random.Random(4)
    for n_in, n_out in ((0, 32), (1, 64), (135, 200), (136, 137), (137, 500), (1000, 1000)):
        msg = bytes(rng.randrange(256) for _ in range(n_in))
        c = Shake()
        L.shake256_init(C.byref(c))
        half = n_in // 2
        L.shake256_absorb(C.byref(c), u8(msg[:half]), half)
        L.shake256_absorb(C.byref(c), u8(msg[half:]), n_in - half)
        L.shake256_finalize(C.byref(c))
        out = (C.c_uint8 * 7)()
        L.shake256_squeeze(C.byref(c), out, 7)
        rest = (C.c_uint8 * (n_out - 7))()
        L.shake256_squeeze(C.byref(c), rest, n_out - 7)
        assert bytes(out) + bytes(rest) == hashlib.shake_256(msg).digest(n_out)


def test_merlin_conformance(kat):
    L = O.lib()
    k = kat["merlin_equivalence_simple"]
    t = Merlin()
    L.merlin_init(C.byref(t), u8(k["protocol"].encode()), len(k["protocol"]))
    L.merlin_append_message(C.byref(t), k["label"].encode(), u8(k["data"].encode()), len(k["data"]))
    out = (C.c_uint8 * 32)()
    L.merlin_challenge_bytes(C.byref(t), k["challenge_label"].encode(), out, 32)
    assert bytes(out).hex() == k["challenge32"]


def test_gens_and_commit_vs_model():
    """MultiCommitGens::new and commit vs the Python model (commitments.rs:20-38,85-98)."""
    L = O.lib()
    label = b"gens_r1cs_sat"
    n = 5
    gens = (Ge * (n + 1))()
    L.oracle_gens_new(gens, n, u8(label), len(label))
    B = G.basepoint()
    stream = hashlib.shake_256(label + B.encode()).digest(64 * (n + 1))
    py = [G.from_uniform_bytes(stream[64 * i:64 * i + 64]) for i in range(n + 1)]
    for i in range(n + 1):
        assert comp(L, gens[i]) == py[i].encode()
    rng = random.Random(5)
    vals = [rng.randrange(M.Q) for _ in range(n)]
    blind = rng.randrange(M.Q)
    v = M.ints_to_table(vals)
    bl = M.ints_to_table([blind])
    out = Ge()
    L.oracle_commit(C.byref(out), O.ptr(v), n, O.ptr(bl), gens, C.byref(gens[n]))
    exp = blind * py[n]
    for x, g in zip(vals, py[:n]):
        exp = exp + x * g
    assert comp(L, out) == exp.encode()


@pytest.mark.parametrize("n", [1, 7, 8, 33, 600])
def test_msm_pippenger_vs_naive(n):
    L = O.lib()
    rng = random.Random(n)
    gens = (Ge * (n + 1))()
    L.oracle_gens_new(gens, n, u8(b"msm-test"), 8)
    vals = []
    for i in range(n):
        kind = rng.random()
        vals.append(0 if kind < 0.2 else 1 if kind < 0.3 else rng.randrange(2**16) if kind < 0.4
                    else M.Q - 1 if kind < 0.45 else rng.randrange(M.Q))
    v = M.ints_to_table(vals)
    out = Ge()
    L.ge_msm(C.byref(out), O.ptr(v), gens, n)
    acc = Ge()
    L.ge_identity(C.byref(acc))
    for i in range(n):
        t = Ge()
        L.ge_scalarmul_bytes(C.byref(t), u8(vals[i].to_bytes(32, "little")), C.byref(gens[i]))
        L.ge_add(C.byref(acc), C.byref(acc), C.byref(t))
    assert comp(L, out) == comp(L, acc)
    if n <= 8:
        pts = [G.decode(comp(L, gens[i])) for i in range(n)]
        exp = G.Pt.identity()
        for x, g in zip(vals, pts):
            exp = exp + x * g
        assert comp(L, out) == exp.encode()


def test_hyrax_commit_rows():
    L = O.lib()
    rng = random.Random(11)
    Ls, Rs = 4, 8
    gens = (Ge * (Rs + 1))()
    L.oracle_gens_new(gens, Rs, u8(b"gens_r1cs_sat"), 13)
    Z = M.ints_to_table([rng.randrange(M.Q) if rng.random() < 0.7 else 0 for _ in range(Ls * Rs)])
    blinds = M.ints_to_table([rng.randrange(M.Q) for _ in range(Ls)])
    out = np.zeros((Ls, 32), dtype=np.uint8)
    L.oracle_hyrax_commit(out.ctypes.data_as(C.c_void_p), O.ptr(Z), Ls, Rs, O.ptr(blinds), gens,
                          C.byref(gens[Rs]), 2)
    for i in range(Ls):
        row = np.ascontiguousarray(Z[i * Rs:(i + 1) * Rs])
        c = Ge()
        L.oracle_commit(C.byref(c), O.ptr(row), Rs, O.ptr(np.ascontiguousarray(blinds[i:i + 1])), gens,
                        C.byref(gens[Rs]))
        assert bytes(out[i]) == comp(L, c)
