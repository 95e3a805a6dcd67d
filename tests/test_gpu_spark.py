"""GPU parity for the SPARK half: SNARK::encode's commitment and the whole SNARK (sat proof +
inst_evals + R1CSEvalProof) produced through the C ABI must be byte-identical to the CPU oracle's
for the same instance and RandomTape seeds, and the oracle's verifier must accept them."""
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
    got = ctx.snark_prove(inst, *seeds)
    exp = O.snark_prove(inst, *seeds)
    assert len(exp["proof"]) > 0
    assert got["comm"] == exp["comm"]
    assert np.array_equal(got["comm_para"], exp["comm_para"])
    assert np.array_equal(got["comm_input"], exp["comm_input"])
    if got["proof"] != exp["proof"]:
        n = min(len(got["proof"]), len(exp["proof"]))
        first = next((i for i in range(n) if got["proof"][i] != exp["proof"][i]), n)
        sat_len = len(O.sat_prove(inst, *seeds)["proof"])
        raise AssertionError(f"SNARK bytes differ at {first} (sat part is {sat_len} bytes; lengths "
                             f"{len(got['proof'])} vs {len(exp['proof'])})")
    assert O.snark_verify(inst, got) == 1
    return got


def test_encode_commitment_matches(ctx):
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3)))
    decomm, comm = ctx.spark_encode(inst)
    decomm.free()
    assert comm == O.snark_prove(inst, SEED_C, SEED_P)["comm"]


def test_point_add_snark(ctx):
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3)))
    check(ctx, inst)


def test_point_add_64_snark(ctx):
    """64 additions: N = 2^10, so the ops forest has both host-proved and device-proved layers"""
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E + 5, 64, rz_one_every=3)))
    check(ctx, inst)


def test_point_mult_snark(ctx):
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 2, 1, weights=[(1 << 127) + 12345])))
    got = check(ctx, inst)
    got2 = ctx.snark_prove(inst, SEED_C, bytes(64))
    assert got2["proof"] != got["proof"] and got2["comm"] == got["comm"]
    assert O.snark_verify(inst, got2) == 1


def test_point_mult_two_ops_snark(ctx):
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 3, 2, weights=[5, (1 << 200) + 77])))
    check(ctx, inst)


def test_product_verifier_accepts_and_rejects(ctx):
    """vpin_snark_verify / vpin_sat_verify (my_lib_verify / my_r1csproof_verify in the product) accept the
    library's and the oracle's proofs and reject tampering in every section, like the oracle's verifier"""
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E + 5, 64, rz_one_every=3)))
    got = ctx.snark_prove(inst, SEED_C, SEED_P)
    assert ctx.snark_verify(inst, got)
    exp = O.snark_prove(inst, SEED_C, SEED_P)
    assert ctx.snark_verify(inst, exp)
    sat = ctx.sat_prove(inst, SEED_C, SEED_P)
    assert ctx.sat_verify(inst, sat)
    n_sat, p = len(sat["proof"]), got["proof"]
    for pos in (10, n_sat // 2, n_sat + 5, n_sat + 96 + 20, n_sat + 96 + 8 + 32 * 8 + 40, (n_sat + len(p)) // 2, len(p) - 2000, len(p) - 40):
        bad = bytearray(p)
        bad[pos] ^= 1
        assert not ctx.snark_verify(inst, got, proof=bytes(bad)), pos
        assert O.snark_verify(inst, got, proof=bytes(bad)) == 0, pos
    badc = bytearray(got["comm"])
    badc[-7] ^= 1
    assert not ctx.snark_verify(inst, got, comm=bytes(badc))
    assert not ctx.snark_verify(inst, got, proof=p[:-1])
    bad = bytearray(sat["proof"])
    bad[len(bad) // 3] ^= 4
    assert not ctx.sat_verify(inst, sat, proof=bytes(bad))
    other = dict(got)
    cp = got["comm_para"].copy()
    cp[0], cp[1] = got["comm_para"][1].copy(), got["comm_para"][0].copy()
    other["comm_para"] = cp
    assert not ctx.snark_verify(inst, other)


def test_product_verifier_mult(ctx):
    inst = GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 2, 1, weights=[(1 << 127) + 12345])))
    got = ctx.snark_prove(inst, SEED_C, SEED_P)
    assert ctx.snark_verify(inst, got)
    assert ctx.snark_verify(inst, O.snark_prove(inst, SEED_C, bytes(64)))


def test_mid_size_snark_verifies_with_both_verifiers(ctx):
    """64 point-mults (221,696 constraints, N = 2^19): the sizes at which the round kernels run multi-block
    with several pairs per thread, the forests have 19 layers and the SPARK generators use wide windows.
    The proof must pass the product's verifier AND the oracle's (independent code), and tampering must fail."""
    from vpin_amd import gadgets as G
    inst = G.synthetic_mult_instance("A", 64)
    d = inst.as_dict()
    inst.free()
    got = ctx.snark_prove(d, SEED_C, SEED_P)
    assert ctx.snark_verify(d, got)
    assert O.snark_verify(d, got) == 1
    bad = bytearray(got["proof"])
    bad[len(bad) * 2 // 3] ^= 0x10
    assert not ctx.snark_verify(d, got, proof=bytes(bad))


def test_every_region_of_the_snark_is_checked(ctx):
    """One flipped bit every 211 bytes of the whole SNARK (sat proof, inst_evals, derefs commitment, the two batched
    product-circuit proofs, the hash-layer claims, the three evaluation proofs) is rejected by the product's verifier;
    a tenth of the positions also go through the oracle's verifier"""
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E + 11, 40, rz_one_every=5)))
    got = ctx.snark_prove(inst, SEED_C, SEED_P)
    assert ctx.snark_verify(inst, got)
    p = got["proof"]
    positions = list(range(5, len(p), 211))
    assert len(positions) > 150
    for k, pos in enumerate(positions):
        bad = bytearray(p)
        bad[pos] ^= 1 << (k % 8)
        assert not ctx.snark_verify(inst, got, proof=bytes(bad)), pos
        if k % 10 == 0:
            assert O.snark_verify(inst, got, proof=bytes(bad)) == 0, pos


@pytest.mark.gpu
@pytest.mark.parametrize("wgs", ["2", "8"])
def test_tail_rounds_on_several_workgroups_per_circuit_give_the_same_bytes(wgs):
    """VPIN_SPARK_TAIL_WGS (round 6, off by default: measured slower -- profiles/r06_ab_tail_wgs.txt): the rounds between 1024 and
    8192 pairs per circuit inside the resident tail kernel, on up to 8 workgroups per circuit (classes of the pair index, partial
    sums added by the last arrival).  Whole SNARKs of conv f=3 and CNN A (2^16 / 2^20 constraints) against the oracle's digests."""
    import hashlib
    import json
    import os
    import vpin_amd
    from vpin_amd import gadgets as G
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        gold = json.load(f)["cases"]
    seed_c, seed_p = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
    os.environ["VPIN_SPARK_TAIL_WGS"] = wgs
    try:
        with vpin_amd.Context(0) as ctx:
            for key in ("3_32-mult", "A-add", "A-mult"):
                g = gold[key]
                inp = G.synthetic_mult_inputs(g["label"]) if g["kind"] == "mult" else G.synthetic_add_inputs(g["label"])
                d = ctx.gadget_point_mult_dev(*inp) if g["kind"] == "mult" else ctx.gadget_point_add_dev(*inp)
                try:
                    res = d.snark_prove(seed_c, seed_p)
                finally:
                    d.free()
                assert hashlib.sha256(res["proof"]).hexdigest() == g["snark_sha256"], key
    finally:
        del os.environ["VPIN_SPARK_TAIL_WGS"]
