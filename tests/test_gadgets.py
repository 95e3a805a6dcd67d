"""The product's C++ gadget builders / synthetic witness generator against the independent Python
model (tests/gadgets_model.py): identical (A,B,C) triplets, assignments and instance padding."""
import numpy as np
import pytest

import gadgets_model as GM
import oracle_lib as O
import pymodel as M
from vpin_amd import gadgets as G


def bytes32(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def same_instance(d, m):
    for k in ("num_cons", "num_vars", "num_inputs", "num_cons_unpadded", "num_vars_unpadded"):
        assert d[k] == m[k], k
    for name in "ABC":
        for a, b in zip(d[name], m[name]):
            assert np.array_equal(a, b), name
    for k in ("vars_para", "vars_input", "vars", "inputs"):
        assert np.array_equal(d[k], m[k]), k


def test_synthetic_points_match_model():
    x, y = G.synthetic_points(0x5650494E, 5)
    exp = GM.synthetic_points(0x5650494E, 5)
    for i, (ex, ey) in enumerate(exp):
        assert int.from_bytes(bytes(x[i]), "little") == ex and int.from_bytes(bytes(y[i]), "little") == ey
        assert (ey * ey - (ex ** 3 + GM.E2_A * ex + GM.E2_B)) % GM.Q == 0  # on E2


def test_point_add_matches_model():
    ops = GM.synthetic_add_ops(77, 9, rz_one_every=4)
    inst = G.point_add(bytes32([o[0] for o in ops]), bytes32([o[1] for o in ops]), bytes32([o[2] for o in ops]),
                       bytes32([o[3] for o in ops]), np.array([o[4] for o in ops], dtype=np.uint8))
    assert inst.is_sat()
    same_instance(inst.as_dict(), GM.instance_new(GM.build_point_add(ops)))
    assert inst.nnz == [16 * 9, 14 * 9, 10 * 9]


@pytest.mark.parametrize("weights", [None, [0, 1, 2, 3]])
def test_point_mult_matches_model(weights):
    ops = GM.synthetic_mult_ops(78, 4 if weights else 2, weights=weights)
    inst = G.point_mult([o[0] for o in ops], bytes32([o[1] for o in ops]), bytes32([o[2] for o in ops]))
    assert inst.is_sat()
    d = inst.as_dict()
    same_instance(d, GM.instance_new(GM.build_point_mult(ops)))
    assert inst.nnz == [5260 * len(ops), 4488 * len(ops), 3201 * len(ops)]
    assert O.is_sat(d) == 1  # the oracle agrees the instance is satisfied


def test_config_shapes():
    """BASELINE.md section 2 / SURVEY.md 8(d): conv f=3 (label 3_32) instance sizes"""
    m = G.synthetic_mult_instance("3_32")
    assert (m.num_cons_unpadded, m.num_vars_unpadded, m.num_cons, m.num_vars) == (62352, 62389, 1 << 16, 1 << 16)
    assert m.is_sat()
    a = G.synthetic_add_instance("3_32")
    assert (a.num_cons_unpadded, a.num_vars_unpadded, a.num_cons, a.num_vars) == (160, 241, 256, 256)
    assert a.is_sat()


def test_unsatisfied_detected():
    ops = GM.synthetic_add_ops(5, 2)
    bad_py = bytes32([o[1] + 1 for o in ops])  # not the point's y: constraint system breaks
    inst = G.point_add(bytes32([o[0] for o in ops]), bad_py, bytes32([o[2] for o in ops]),
                       bytes32([o[3] for o in ops]), np.zeros(2, dtype=np.uint8))
    # the add gadget only constrains the formulas, which still hold for any (px,py): stays satisfied
    assert inst.is_sat()
    d = inst.as_dict()
    d["vars"][6, 0] ^= np.uint64(1)
    assert O.is_sat(d) == 0
