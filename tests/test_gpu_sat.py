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
