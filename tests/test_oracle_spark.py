"""The oracle's SPARK half (SNARK::encode + R1CSEvalProof): the restated verifier accepts the restated
prover on both gadget kinds, the SNARK's sat prefix is exactly the stand-alone sat proof, tampering
anywhere in the evaluation proof or in the computation commitment is rejected, and proofs are a
deterministic function of the injected seeds."""
import numpy as np
import pytest

import gadgets_model as GM
import oracle_lib as O

SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))


@pytest.fixture(scope="module")
def add_inst():
    return GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3)))


@pytest.fixture(scope="module")
def mult_inst():
    return GM.instance_new(GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 1, 1, weights=[(1 << 127) + 12345])))


@pytest.fixture(scope="module")
def add_res(add_inst):
    return O.snark_prove(add_inst, SEED_C, SEED_P)


def test_add_snark_roundtrip(add_inst, add_res):
    assert len(add_res["proof"]) > 0
    assert O.snark_verify(add_inst, add_res) == 1
    again = O.snark_prove(add_inst, SEED_C, SEED_P)
    assert again["proof"] == add_res["proof"] and again["comm"] == add_res["comm"]


def test_snark_prefix_is_the_sat_proof(add_inst, add_res):
    sat = O.sat_prove(add_inst, SEED_C, SEED_P)
    n = len(sat["proof"])
    assert add_res["proof"][:n] == sat["proof"]
    # SNARK.inst_evals follow as three Montgomery-form scalars (lib.rs:334-338)
    evals = np.frombuffer(add_res["proof"][n:n + 96], dtype=np.uint64).reshape(3, 4)
    assert (evals == sat["inst_evals"]).all()


def test_comm_layout(add_inst, add_res):
    hdr = np.frombuffer(add_res["comm"][:48], dtype=np.uint64)
    nnz_max = max(len(add_inst[k][0]) for k in "ABC")
    N = 1 << (nnz_max - 1).bit_length()
    nx, ny = O.log2(add_inst["num_cons"]), O.log2(2 * add_inst["num_vars"])
    assert list(hdr) == [add_inst["num_cons"], add_inst["num_vars"], add_inst["num_inputs"], 3, N, 1 << max(nx, ny)]


def test_tamper_rejected(add_inst, add_res):
    sat_len = len(O.sat_prove(add_inst, SEED_C, SEED_P)["proof"])
    p = add_res["proof"]
    # inst_evals, comm_derefs, product-layer claims, sum-check polys, hash-layer evals, the last PolyEvalProof
    for pos in (sat_len + 5, sat_len + 96 + 20, sat_len + 96 + 8 + 32 * 8 + 40, (sat_len + len(p)) // 2,
                len(p) - 2000, len(p) - 40):
        bad = bytearray(p)
        bad[pos] ^= 1
        assert O.snark_verify(add_inst, add_res, proof=bytes(bad)) == 0, pos
    badc = bytearray(add_res["comm"])
    badc[-7] ^= 1
    assert O.snark_verify(add_inst, add_res, comm=bytes(badc)) == 0
    assert O.snark_verify(add_inst, add_res, proof=p[:-1]) == 0


def test_wrong_witness_commitment_rejected(add_inst, add_res):
    other = dict(add_res)
    cp = add_res["comm_para"].copy()
    cp[0], cp[1] = add_res["comm_para"][1].copy(), add_res["comm_para"][0].copy()
    other["comm_para"] = cp
    assert O.snark_verify(add_inst, other) == 0


def test_mult_snark_roundtrip(mult_inst):
    res = O.snark_prove(mult_inst, SEED_C, SEED_P)
    assert len(res["proof"]) > 0
    assert O.snark_verify(mult_inst, res) == 1
    res2 = O.snark_prove(mult_inst, SEED_C, bytes(64))
    assert res2["proof"] != res["proof"] and O.snark_verify(mult_inst, res2) == 1
    t = O.spark_timings()
    assert t["total"] >= t["sat"] > 0
