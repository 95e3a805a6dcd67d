"""GPU twins of tests/test_reference_pins.py: the product on the Spartan README profile's shape (2^20 constraints /
variables / non-zeros per matrix, 10 inputs -- a synthetic, non-gadget R1CS through the host-buffer entry point
vpin_snark_prove) produces the oracle's bytes and the lengths Spartan's own profiler prints (Spartan/README.md:363,372,375);
and the device-built gadget instances have the non-zero counts the reference's hand-tuned generator sizes assume
(point_mult.rs:27-67, point_addition.rs:38-70) for every configuration, the 6000-operation one included."""
import os

import pytest

import bincode_layout as BL
import oracle_lib as O
import test_reference_pins as RP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


def test_readme_shape_snark_bytes_and_lengths(ctx):
    ref = RP.PINS["spartan_readme_profile"]
    inst = RP.readme_shape_instance(20, ref["number_of_inputs"])
    got = ctx.snark_prove(inst, RP.SEED_C, RP.SEED_P)
    sec = BL.snark_sections(got["proof"])
    assert sec["len_r1cs_sat_proof"] == ref["len_r1cs_sat_proof"]
    assert sec["len_product_layer_proof"] == ref["len_product_layer_proof"]
    assert sec["len_r1cs_eval_proof"] == ref["len_r1cs_eval_proof"]
    exp = O.snark_prove(inst, RP.SEED_C, RP.SEED_P, threads=os.cpu_count() or 8)
    assert got["comm"] == exp["comm"]
    assert got["proof"] == exp["proof"]
    assert O.snark_verify(inst, got) == 1
    assert ctx.snark_verify(inst, got)


def test_device_instances_have_the_nnz_the_reference_sizes_for(ctx):
    from vpin_amd import gadgets as G
    seen = 0
    for label, cfg in G.CONFIGS.items():
        if cfg["n_mult"]:
            g = ctx.gadget_point_mult_dev(*G.synthetic_mult_inputs(label))
            actual = max(g.nnz)
            g.free()
            d = RP.mult_declared(cfg["n_mult"])
            assert RP.BL_next_pow2(d) == RP.BL_next_pow2(actual), (label, "mult", cfg["n_mult"], d, actual)
            seen += 1
        g = ctx.gadget_point_add_dev(*G.synthetic_add_inputs(label))
        actual = max(g.nnz)
        g.free()
        d = RP.add_declared(cfg["n_add"])
        assert RP.BL_next_pow2(d) == RP.BL_next_pow2(actual), (label, "add", cfg["n_add"], d, actual)
        seen += 1
    assert seen == 20
