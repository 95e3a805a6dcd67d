"""Pins to numbers the REFERENCE holds (tests/golden/reference_pins.json), beyond the F_q / unipoly / MLE KATs of
test_oracle_fq.py / test_oracle_poly.py:

* the proof lengths Spartan's own profiler prints for its 2^20 instance (Spartan/README.md:363,372,375) -- pins the
  oracle's bincode layout, the number of rounds / layers / commitments of every sub-proof and the generator sizing;
* the hand-tuned `num_non_zero_entries` tables of point_mult.rs:27-67 / point_addition.rs:38-70: the reference sizes its
  SPARK generators from them and its commit asserts that they fit (commitments.rs:95), so for every operation count of
  BASELINE.json's configurations they must round to the same power of two as the real maximum nnz of the gadget
  instance -- pins the constraint counts of the product's gadget builders (and of tests/gadgets_model.py);
* SURVEY.md A.1's challenge_scalar usage vector on both Merlin implementations (oracle/keccak.c and the product's
  host/transcript.h).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bincode_layout as BL
import oracle_lib as O
import pymodel as M

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "reference_pins.json")))

SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))


def readme_shape_instance(log_n=20, num_inputs=10):
    """A satisfiable instance of the README profile's shape: num_cons = num_vars = nnz(A) = nnz(B) = nnz(C) = 2^log_n.
    Constraint i: (1 * z_i) * (v_i * z_i) = (v_i^2-free form) -> A[i,i] = 1, B[i,i] = 1, C[i,i] = z_i's value, with
    z_i = vars_i, i.e. vars_i * vars_i = vars_i * vars_i.  Proof LENGTHS depend on the shape only."""
    n = 1 << log_n
    small = [(k % 251) + 2 for k in range(n)]
    lut = {v: M.to_mont_limbs(v) for v in set(small)}
    vals = np.array([lut[v] for v in small], dtype=np.uint64)
    one = np.tile(np.array(M.to_mont_limbs(1), dtype=np.uint64), (n, 1))
    idx = np.arange(n, dtype=np.uint32)
    inputs = np.array([M.to_mont_limbs(3 * k + 1) for k in range(num_inputs)], dtype=np.uint64).reshape(num_inputs, 4)
    zeros = np.zeros((n, 4), dtype=np.uint64)
    return dict(num_cons=n, num_vars=n, num_inputs=num_inputs, num_cons_unpadded=n, num_vars_unpadded=n,
                A=(idx, idx, one), B=(idx, idx, one), C=(idx, idx, vals),
                vars_para=vals, vars_input=zeros, vars=vals.copy(), inputs=inputs)


def test_readme_shape_is_satisfiable_small():
    inst = readme_shape_instance(8, 3)
    assert O.is_sat(inst) == 1


def test_oracle_proof_lengths_equal_spartan_readme_profile():
    """Spartan/README.md:344-377"""
    ref = PINS["spartan_readme_profile"]
    inst = readme_shape_instance(20, ref["number_of_inputs"])
    assert inst["num_cons"] == ref["number_of_constraints"] and inst["num_vars"] == ref["number_of_variables"]
    assert all(len(inst[k][0]) == ref["number_non_zero_entries"] for k in "ABC")
    res = O.snark_prove(inst, SEED_C, SEED_P, threads=os.cpu_count() or 8)
    sec = BL.snark_sections(res["proof"])
    assert sec["len_r1cs_sat_proof"] == ref["len_r1cs_sat_proof"]
    assert sec["len_product_layer_proof"] == ref["len_product_layer_proof"]
    assert sec["len_r1cs_eval_proof"] == ref["len_r1cs_eval_proof"]
    assert sec["total"] == ref["len_r1cs_sat_proof"] + 96 + ref["len_r1cs_eval_proof"]
    # the sat half stands alone (SNARK's first field, lib.rs:330-338) and has the same length
    sat = O.sat_prove(inst, SEED_C, SEED_P, threads=os.cpu_count() or 8)
    assert len(sat["proof"]) == ref["len_r1cs_sat_proof"] and res["proof"][:len(sat["proof"])] == sat["proof"]


def test_layout_walk_structure_small():
    """the walk recovers the structure the protocol dictates on a small gadget instance"""
    import gadgets_model as GM
    inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3)))
    res = O.snark_prove(inst, SEED_C, SEED_P)
    sec = BL.snark_sections(res["proof"])
    t = sec["tree"]
    nx, ny = O.log2(inst["num_cons"]), O.log2(2 * inst["num_vars"])
    sat = t["r1cs_sat_proof"]
    assert sat["sc_proof_phase1"]["comm_polys"].count == nx and sat["sc_proof_phase2"]["comm_polys"].count == ny
    assert sat["comm_vars"]["C"].count == 1 << (O.log2(inst["num_vars"]) // 2)
    N = max(BL_next_pow2(len(inst[k][0])) for k in "ABC")
    Mm = max(inst["num_cons"], 2 * inst["num_vars"])
    prod = t["r1cs_eval_proof"]["proof"]["poly_eval_network_proof"]["proof_prod_layer"]
    assert prod["proof_ops"]["proof"].count == O.log2(N) and prod["proof_mem"]["proof"].count == O.log2(Mm)
    assert [prod["proof_ops"]["claims_dotp"][k].count for k in ("left", "right", "weight")] == [6, 6, 6]
    assert [prod["proof_mem"]["claims_dotp"][k].count for k in ("left", "right", "weight")] == [0, 0, 0]


def BL_next_pow2(x):
    p = 1
    while p < x:
        p *= 2
    return p


# ---- the num_non_zero_entries tables --------------------------------------------------------------------------------

def _declared(rules, N, formula):
    for r in rules:
        if eval(r["when"], {"N": N}):
            return formula(*r["params"])
    raise AssertionError("no rule matched")


def mult_declared(N):
    t = PINS["point_mult_nnz_params"]
    n = t["n"]
    return _declared(t["rules"], N, lambda p1, p2, p3: p1 * ((p2 * n) + (p3 * N)))


def add_declared(N):
    t = PINS["point_add_nnz_params"]
    return _declared(t["rules"], N, lambda p1, p2, p3: p1 * (p2 // p3) * N)


def _product_nnz(kind, N):
    """max nnz over A, B, C of the PRODUCT's host-built instance (vpin_gadget_point_*; no GPU involved)"""
    from vpin_amd import gadgets as G
    inst = G.synthetic_mult_instance("A", N) if kind == "mult" else G.synthetic_add_instance("A", N)
    try:
        return list(inst.nnz)
    finally:
        inst.free()


@pytest.fixture(scope="module")
def nnz_model():
    """nnz(N) of the product's builders is affine in the operation count (N shifted copies of a template plus the
    special columns): fit on N = 1, 2 and confirm on a third size before extrapolating to the configurations."""
    out = {}
    for kind, probe in (("mult", 7), ("add", 37)):
        a1, a2, a3 = _product_nnz(kind, 1), _product_nnz(kind, 2), _product_nnz(kind, probe)
        slope = [a2[m] - a1[m] for m in range(3)]
        icpt = [a1[m] - slope[m] for m in range(3)]
        assert [icpt[m] + slope[m] * probe for m in range(3)] == a3
        out[kind] = (slope, icpt)
    return out


def test_declared_nnz_rounds_like_actual_nnz_for_every_config(nnz_model):
    from vpin_amd import gadgets as G
    seen = 0
    for label, cfg in G.CONFIGS.items():
        for kind, N, declared in (("mult", cfg["n_mult"], mult_declared), ("add", cfg["n_add"], add_declared)):
            if N == 0:
                continue
            slope, icpt = nnz_model[kind]
            actual = max(icpt[m] + slope[m] * N for m in range(3))
            d = declared(N)
            assert BL_next_pow2(d) == BL_next_pow2(actual), (label, kind, N, d, actual)
            seen += 1
    assert seen == 20  # 9 point-mult + 11 point-add instances


def test_python_gadget_model_has_the_same_nnz():
    import gadgets_model as GM
    g = GM.build_point_mult(GM.synthetic_mult_ops(2, 2))
    assert [len(g[k]) for k in "ABC"] == _product_nnz("mult", 2)
    g = GM.build_point_add(GM.synthetic_add_ops(1, 5))
    assert [len(g[k]) for k in "ABC"] == _product_nnz("add", 5)


# ---- challenge_scalar usage KAT ---------------------------------------------------------------------------------------

Q = 2**252 + 27742317777372353535851937790883648493


def test_challenge_scalar_kat_oracle_and_product():
    k = PINS["challenge_scalar_usage_kat"]
    # oracle/keccak.c
    L = O.lib()
    m = O.Merlin()
    lab = k["transcript_label"].encode()
    L.merlin_init(C.byref(m), lab, len(lab))
    msg = k["append_message"].encode()
    L.merlin_append_message(C.byref(m), k["append_label"].encode(), msg, len(msg))
    wide = (C.c_uint8 * 64)()
    L.merlin_challenge_bytes(C.byref(m), k["challenge_label"].encode(), wide, 64)
    wide_o = bytes(wide)
    wide2 = (C.c_uint8 * 64)()
    L.merlin_challenge_bytes(C.byref(m), k["challenge_label"].encode(), wide2, 64)
    # the product's host transcript (vpin_host_merlin_kat: new, append_message, challenge_bytes)
    import vpin_amd
    out = (C.c_uint8 * 64)()
    rc = vpin_amd.lib().vpin_host_merlin_kat(lab, k["append_label"].encode(), (C.c_uint8 * len(msg))(*msg), len(msg),
                                             k["challenge_label"].encode(), out, 64)
    assert rc == 0
    wide_p = bytes(out)
    assert wide_o == wide_p
    assert wide_o.hex().startswith(k["wide_bytes_prefix"]) and wide_o.hex().endswith(k["wide_bytes_suffix"])
    s = int.from_bytes(wide_o, "little") % Q
    assert s.to_bytes(32, "little").hex() == k["scalar_canonical_le"]
    assert ((s << 256) % Q).to_bytes(32, "little").hex() == k["scalar_montgomery_le"]
    # the oracle's from_bytes_wide (ristretto255.rs:442-473) lands on the same Montgomery limbs
    f = L.fq_from_bytes_wide(wide)
    assert b"".join(int(x).to_bytes(8, "little") for x in f.limbs()).hex() == k["scalar_montgomery_le"]
    s2 = (int.from_bytes(bytes(wide2), "little") % Q).to_bytes(32, "little").hex()
    assert s2.startswith(k["next_scalar_canonical_prefix"]) and s2.endswith(k["next_scalar_canonical_suffix"])
