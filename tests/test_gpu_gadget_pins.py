"""GPU half of the reference-text gadget pins (tests/test_gadget_pins.py has the story): the instances the DEVICE
builders produce (vpin_gadget_point_{add,mult}_dev: template replication, witness synthesis kernels, the fast
Jacobian path and the step-by-step fallback on vanishing denominators) are read back through the C ABI
(vpin_dev_instance_triplets, the three assignment tables, the public input) and must hash to what the
reference's own statements produced (tests/golden/gadget_pins.json <- tests/golden/make_gadget_pins.py)."""
import numpy as np
import pytest

import test_gadget_pins as TP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


def dev_dict(g):
    d = dict(num_cons=g.num_cons, num_vars=g.num_vars, num_inputs=g.num_inputs,
             num_cons_unpadded=g.num_cons_unpadded, num_vars_unpadded=g.num_vars_unpadded)
    for m, k in enumerate("ABC"):
        d[k] = g.triplets(m)
    for k in ("vars_para", "vars_input", "vars"):
        d[k] = getattr(g, k).read()
    d["inputs"] = g.inputs
    return d


@pytest.mark.parametrize("case", TP.PINS["mult"], ids=lambda c: c["name"])
def test_device_builder_point_mult_equals_reference_text(ctx, case):
    g = ctx.gadget_point_mult_dev([int(w) for w in case["weights"]], TP.u8(case["px"]), TP.u8(case["py"]))
    try:
        TP.check_instance_dict(dev_dict(g), case)
        assert g.is_sat() or case["name"].endswith("edge")
    finally:
        g.free()


@pytest.mark.parametrize("case", TP.PINS["add"], ids=lambda c: c["name"])
def test_device_builder_point_add_equals_reference_text(ctx, case):
    g = ctx.gadget_point_add_dev(TP.u8(case["px"]), TP.u8(case["py"]), TP.u8(case["rx"]), TP.u8(case["ry"]),
                                 np.array(case["rz"], dtype=np.uint8))
    try:
        TP.check_instance_dict(dev_dict(g), case)
        assert g.is_sat() or case["name"].endswith("edge")
    finally:
        g.free()
