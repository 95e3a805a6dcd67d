"""GPU parity for the whole hot path: the sat proof produced by the HIP path (vpin_sat_prove
through the C ABI) must be byte-identical to the CPU oracle's proof for the same instance and
the same injected RandomTape seeds, and the oracle's verifier must accept it."""
import numpy as np
import pytest

import gadgets_model as GM
import oracle_lib as O

pytestmark = pytest.mark.gpu

SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


def check(ctx, inst, seeds=(SEED_C, SEED_P)):
    got = ctx.sat_prove(inst, *seeds)
    exp = O.sat_prove(inst, *seeds)
    assert len(exp["proof"]) > 0
    assert np.array_equal(got["comm_para"], exp["comm_para"])
    assert np.array_equal(got["comm_input"], exp["comm_input"])
    assert np.array_equal(got["rx"], exp["rx"])
    assert np.array_equal(got["ry"], exp["ry"])
    assert np.array_equal(got["inst_evals"], exp["inst_evals"])
    assert got["proof"] == exp["proof"]
    assert O.sat_verify(inst, got) == 1
    return got


def test_point_add_instance(ctx):
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3)))
    check(ctx, inst)


def test_point_add_16_conv3_shape(ctx):
    """config 1 (conv f=3): 16 point additions -> 160 constraints / 2^8 padded"""
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E + 1, 16, rz_one_every=3)))
    assert inst["num_cons"] == 256 and inst["num_vars"] == 256
    check(ctx, inst)


def test_point_mult_instance(ctx):
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 2, 1, weights=[(1 << 127) + 12345])))
    got = check(ctx, inst)
    # a different proof seed changes the proof, not the commitments
    got2 = ctx.sat_prove(inst, SEED_C, bytes(64))
    assert got2["proof"] != got["proof"] and np.array_equal(got2["comm_para"], got["comm_para"])
    assert O.sat_verify(inst, got2) == 1


def test_point_mult_small_weights(ctx):
    """conv-style weights {0,1,2}: exercises the infinity / all-zero-bit paths of the gadget"""
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 3, 3, weights=[0, 1, 2])))
    check(ctx, inst)


def test_r1cs_kernels_vs_oracle(ctx):
    """device SpMV / sparse eval table / matrix evaluation against the oracle's host loops"""
    import ctypes as C
    import pymodel as M
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(31, 2)))
    L = O.lib()
    r = O.make_r1cs(inst)
    nv, nc = inst["num_vars"], inst["num_cons"]
    di = ctx.r1cs_upload(inst)
    tv = ctx.upload(inst["vars"])
    z = ctx.r1cs_build_z(di, tv, inst["inputs"])
    zh = z.read()
    assert np.array_equal(zh[:nv], inst["vars"]) and M.table_to_ints(zh[nv:nv + 3]) == [1, GM.E2_A, 0]
    Az, Bz, Cz = ctx.r1cs_multiply_vec(di, z)
    exp = [np.zeros((nc, 4), dtype=np.uint64) for _ in range(3)]
    L.oracle_r1cs_multiply_vec(C.byref(r), O.ptr(zh), *[O.ptr(e) for e in exp])
    for got, e in zip((Az, Bz, Cz), exp):
        assert np.array_equal(got.read(), e)
    rng = np.random.default_rng(1)
    rx = M.ints_to_table([int(rng.integers(1, 2**62)) ** 4 % M.Q for _ in range(O.log2(nc))])
    ry = M.ints_to_table([int(rng.integers(1, 2**62)) ** 4 % M.Q for _ in range(O.log2(nv) + 1)])
    rabc = M.ints_to_table([int(rng.integers(1, 2**62)) ** 4 % M.Q for _ in range(3)])
    erx, ery = ctx.eq_table(rx), ctx.eq_table(ry)
    tabs = [np.zeros((2 * nv, 4), dtype=np.uint64) for _ in range(3)]
    erx_h = erx.read()
    L.oracle_r1cs_eval_table_sparse(C.byref(r), O.ptr(erx_h), *[O.ptr(t) for t in tabs])
    ints = [M.table_to_ints(t) for t in tabs]
    ra, rb, rc = M.table_to_ints(rabc)
    comb = M.ints_to_table([(ra * a + rb * b + rc * c) % M.Q for a, b, c in zip(*ints)])
    assert np.array_equal(ctx.r1cs_eval_table(di, erx, rabc).read(), comb)
    ev = np.zeros((3, 4), dtype=np.uint64)
    L.oracle_r1cs_evaluate(C.byref(r), O.ptr(rx), O.ptr(ry), O.ptr(ev))
    assert np.array_equal(ctx.r1cs_evaluate(di, erx, ery), ev)
    di.free()


def test_resident_path_same_proof(ctx):
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(9, 5, rz_one_every=2)))
    di = ctx.r1cs_upload(inst)
    tabs = [ctx.upload(inst[k]) for k in ("vars_para", "vars_input", "vars")]
    a = ctx.sat_prove_resident(di, *tabs, inst["inputs"], SEED_C, SEED_P)
    b = ctx.sat_prove(inst, SEED_C, SEED_P)
    assert a["proof"] == b["proof"] and O.sat_verify(inst, a) == 1
    # the assignments must be untouched by the proof (they are inputs, not scratch)
    for t, k in zip(tabs, ("vars_para", "vars_input", "vars")):
        assert np.array_equal(t.read(), inst[k])
    di.free()


def test_unsatisfied_witness_gives_the_reference_bytes(ctx):
    """A witness that does not satisfy the instance: phase 1's claim (0) is then not the true sum, so the prover must
    leave the leading-coefficient rounds (which derive t(1) from the claim) for the three-sum kernels; the proof is
    still byte-identical to the oracle's (and the verifier rejects it)."""
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E + 9, 5)))
    assert O.is_sat(inst)
    bad = dict(inst)
    for k in ("vars_input", "vars"):
        t = np.array(inst[k], copy=True)
        t[13] = t[14]  # x3 of the first addition := y3
        bad[k] = t
    assert not O.is_sat(bad)
    got = ctx.sat_prove(bad, SEED_C, SEED_P)
    exp = O.sat_prove(bad, SEED_C, SEED_P)
    assert got["proof"] == exp["proof"]
    assert O.sat_verify(bad, got) != 1
    # and a mult instance large enough for the multi-workgroup kernels
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 10, 1)))
    bad = dict(inst)
    for k in ("vars_input", "vars"):
        t = np.array(inst[k], copy=True)
        t[130] = t[131]  # dx of the first doubling := dx of the second
        bad[k] = t
    assert not O.is_sat(bad)
    assert ctx.sat_prove(bad, SEED_C, SEED_P)["proof"] == O.sat_prove(bad, SEED_C, SEED_P)["proof"]


def test_every_region_of_the_sat_proof_is_checked(ctx):
    """One flipped bit every 61 bytes of the sat proof -- row commitments, both ZK sum-checks (comm_polys, comm_evals and the
    delta / beta / z / z_delta / z_beta of every round's DotProductProof), the knowledge / product / equality proofs, the
    evaluation proof -- must be rejected by the product's verifier (which collects the rounds' group equations and checks
    them after the transcript pass) exactly as by the oracle's, which checks them in place."""
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E + 9, 24, rz_one_every=4)))
    sat = ctx.sat_prove(inst, SEED_C, SEED_P)
    assert ctx.sat_verify(inst, sat) and O.sat_verify(inst, sat) == 1
    p = sat["proof"]
    positions = list(range(3, len(p), 61))
    assert len(positions) > 100
    for k, pos in enumerate(positions):
        bad = bytearray(p)
        bad[pos] ^= 1 << (k % 8)
        assert not ctx.sat_verify(inst, sat, proof=bytes(bad)), pos
        if k % 7 == 0:  # the oracle's verifier is slower: a seventh of the positions
            assert O.sat_verify(inst, sat, proof=bytes(bad)) == 0, pos
