"""The gadgets pinned to the reference's SOURCE TEXT.

tests/golden/gadget_pins.json is what the statements of vPIN_proof_generation/src/point_mult.rs:20-729 and
point_addition.rs:18-313 produce when they are translated mechanically (tests/golden/make_gadget_pins.py: a
line-by-line Rust-subset -> Python translation executed in the build container; nobody re-typed a
coefficient): the (row, col, value) triplets of A, B, C in push order, the sizes, the declared
num_non_zero_entries and the three assignment vectors, as counts + SHA-256.

Checked here against them, without a GPU:
  * tests/gadgets_model.py -- the hand restatement every config digest of tests/golden/config_digests.json
    was generated through;
  * the product's host builders vpin_gadget_point_{add,mult} (C ABI);
  * the transcribed num_non_zero_entries rules of tests/golden/reference_pins.json.
The device builders are checked in tests/test_gpu_gadget_pins.py."""
import hashlib
import json
import os

import numpy as np
import pytest

import gadgets_model as GM
import pymodel as M

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "gadget_pins.json")) as f:
    PINS = json.load(f)
Q = M.Q


# ---- the serialisation the fixture names ---------------------------------------------------------------------------

def digest_triplets(rows, cols, vals_int):
    h = hashlib.sha256()
    for r, c, v in zip(rows, cols, vals_int):
        h.update(int(r).to_bytes(8, "little") + int(c).to_bytes(8, "little") + int(v).to_bytes(32, "little"))
    return h.hexdigest()


def digest_ints(vals_int):
    h = hashlib.sha256()
    for v in vals_int:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def le(b):
    return int.from_bytes(bytes(b), "little")


def mult_ops(case):
    return [(int(w), le(x), le(y)) for w, x, y in zip(case["weights"], case["px"], case["py"])]


def add_ops(case):
    return [(le(a), le(b), le(c), le(d), int(z)) for a, b, c, d, z in
            zip(case["px"], case["py"], case["rx"], case["ry"], case["rz"])]


def check_model(g, case):
    assert (g["num_cons"], g["num_vars"], g["num_inputs"]) == (case["num_cons"], case["num_vars"], case["num_inputs"])
    assert [len(g[k]) for k in "ABC"] == case["nnz"]
    for k in "ABC":
        t = g[k]
        assert digest_triplets([x[0] for x in t], [x[1] for x in t], [x[2] % Q for x in t]) == case["sha256"][k], k
    for k in ("vars_para", "vars_input", "vars"):
        assert digest_ints([v % Q for v in g[k]]) == case["sha256"][k], k
    assert digest_ints([v % Q for v in g["inputs"]]) == case["sha256"]["inputs"]


@pytest.mark.parametrize("case", PINS["mult"], ids=lambda c: c["name"])
def test_python_model_point_mult_equals_reference_text(case):
    # the model reduces rz / coordinates the way the gadget does (Scalar::from_bytes_mod_order)
    check_model(GM.build_point_mult(mult_ops(case), n=PINS["n"]), case)


@pytest.mark.parametrize("case", PINS["add"], ids=lambda c: c["name"])
def test_python_model_point_add_equals_reference_text(case):
    ops = [(a, b, c, d, 0 if z == 0 else 1) for a, b, c, d, z in add_ops(case)]   # point_addition.rs:189-193
    check_model(GM.build_point_add(ops), case)


# ---- the product's host builders (no GPU) ---------------------------------------------------------------------------

def unmont(vals):
    """(k,4) u64 Montgomery limbs -> ints; the gadgets use a handful of distinct coefficients"""
    cache, out = {}, []
    for row in np.asarray(vals, dtype=np.uint64).reshape(-1, 4):
        key = row.tobytes()
        if key not in cache:
            cache[key] = M.from_mont_limbs(row)
        out.append(cache[key])
    return out


def check_instance_dict(d, case):
    """d: padded instance in the C ABI's layout (Instance::new applied: power-of-two sizes, columns >= num_vars
    moved up by the padding, SP/lib.rs:196-200) -> undo the padding and compare with the reference's own arrays"""
    nvr, nv_pad = d["num_vars_unpadded"], d["num_vars"]
    assert (d["num_cons_unpadded"], nvr, d["num_inputs"]) == (case["num_cons"], case["num_vars"], case["num_inputs"])
    for k in "ABC":
        rows, cols, vals = d[k]
        cols = np.asarray(cols, dtype=np.int64)
        cols = np.where(cols >= nv_pad, cols - (nv_pad - nvr), cols)
        assert len(rows) == case["nnz"]["ABC".index(k)]
        assert digest_triplets(rows, cols, unmont(vals)) == case["sha256"][k], k
    for k in ("vars_para", "vars_input", "vars"):
        tab = np.asarray(d[k]).reshape(-1, 4)
        assert not tab[nvr:].any(), "padding of %s is not zero" % k
        assert digest_ints(unmont(tab[:nvr])) == case["sha256"][k], k
    assert digest_ints(unmont(d["inputs"])) == case["sha256"]["inputs"]


def u8(rows):
    return np.array(rows, dtype=np.uint8).reshape(-1, 32)


@pytest.mark.parametrize("case", PINS["mult"], ids=lambda c: c["name"])
def test_host_builder_point_mult_equals_reference_text(case):
    from vpin_amd import gadgets as G
    inst = G.point_mult([int(w) for w in case["weights"]], u8(case["px"]), u8(case["py"]))
    try:
        check_instance_dict(inst.as_dict(), case)
    finally:
        inst.free()


@pytest.mark.parametrize("case", PINS["add"], ids=lambda c: c["name"])
def test_host_builder_point_add_equals_reference_text(case):
    from vpin_amd import gadgets as G
    inst = G.point_add(u8(case["px"]), u8(case["py"]), u8(case["rx"]), u8(case["ry"]),
                       np.array(case["rz"], dtype=np.uint8))
    try:
        check_instance_dict(inst.as_dict(), case)
    finally:
        inst.free()


# ---- sizes and the declared nnz of every configuration --------------------------------------------------------------

def test_shapes_and_declared_nnz_of_every_config_come_from_the_reference_text():
    """vpin_gadget_shape (what the CLI and the generator-set preparation size everything from) against the sizes the
    reference's own statements give; and the hand-transcribed param rules of reference_pins.json against the
    mechanically evaluated chain, at every operation count of BASELINE's configurations and at the chain's
    branch boundaries."""
    import ctypes as C
    from vpin_amd.capi import lib
    import test_reference_pins as RP
    L = lib()
    L.vpin_gadget_shape.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t * 3]
    for kind, rows, declared in (("mult", PINS["mult_shapes"], RP.mult_declared), ("add", PINS["add_shapes"], RP.add_declared)):
        for r in rows:
            assert declared(r["ops"]) == r["num_non_zero_entries"], (kind, r)
            nc, nv, nnz = C.c_size_t(), C.c_size_t(), (C.c_size_t * 3)()
            assert L.vpin_gadget_shape(1 if kind == "mult" else 0, r["ops"], C.byref(nc), C.byref(nv), nnz) == 0
            assert (nc.value, nv.value) == (GM.next_pow2(r["num_cons"]), GM.next_pow2(max(r["num_vars"], r["num_inputs"] + 1))), (kind, r)
    from vpin_amd import gadgets as G
    have_m = {r["ops"] for r in PINS["mult_shapes"]}
    have_a = {r["ops"] for r in PINS["add_shapes"]}
    for label, cfg in G.CONFIGS.items():
        assert cfg["n_mult"] == 0 or cfg["n_mult"] in have_m, label
        assert cfg["n_add"] in have_a, label
