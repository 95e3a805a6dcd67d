"""Regenerate tests/golden/*.json from the Python big-integer models and the CPU oracle.
Run from the repo root: python tests/golden/make_golden.py
(The reference itself cannot produce vectors here: it is Rust and no toolchain exists)"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gadgets_model as GM  # noqa: E402
import oracle_lib as O  # noqa: E402
import pymodel as M  # noqa: E402

Q = M.Q


def sumcheck_vectors():
    """seeded tables -> per-round (e0,e2,e3)/(e0,e2) and folded tables, from the big-int model"""
    rng = random.Random(20240601)
    out = {"_source": "tests/pymodel.py (Python big integers) restating Spartan/src/sumcheck.rs:460-469,624-652 and "
                      "dense_mlpoly.rs:78-94,229-236; canonical integers in hex"}
    n = 32
    tabs = [[rng.randrange(Q) if rng.random() > 0.25 else 0 for _ in range(n)] for _ in range(4)]
    rs = [rng.randrange(Q) for _ in range(5)]
    rounds = []
    cur = [list(t) for t in tabs]
    for r in rs:
        e = M.sc_cubic_round(*cur)
        eq = M.sc_quad_round(cur[0], cur[1])
        cur = [M.bound_top(t, r) for t in cur]
        rounds.append({"cubic": [hex(x) for x in e], "quad": [hex(x) for x in eq], "r": hex(r),
                       "folded_first": [hex(t[0]) for t in cur]})
    out["tables"] = [[hex(x) for x in t] for t in tabs]
    out["rounds"] = rounds
    r4 = [rng.randrange(Q) for _ in range(4)]
    out["eq"] = {"r": [hex(x) for x in r4], "evals": [hex(x) for x in M.eq_evals(r4)]}
    return out


def proof_vectors():
    """proof digests of small gadget instances under fixed seeds, from the C oracle"""
    seed_c, seed_p = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
    out = {"_source": "oracle/sat.c (C restatement of vPIN's sat prover) on instances from tests/gadgets_model.py; "
                      "seed_commit = bytes(range(64)), seed_proof = bytes((7*i+3)%256)",
           "cases": []}
    cases = [("add6", GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3))),
             ("mult1", GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 2, 1, weights=[(1 << 127) + 12345]))),
             ("mult3_small_weights", GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 3, 3, weights=[0, 1, 2])))]
    for name, g in cases:
        inst = GM.instance_new(g)
        res = O.sat_prove(inst, seed_c, seed_p)
        assert O.sat_verify(inst, res) == 1
        out["cases"].append({"name": name, "num_cons": inst["num_cons"], "num_vars": inst["num_vars"],
                             "proof_len": len(res["proof"]), "proof_sha256": hashlib.sha256(res["proof"]).hexdigest(),
                             "comm_para_sha256": hashlib.sha256(res["comm_para"].tobytes()).hexdigest(),
                             "comm_input_sha256": hashlib.sha256(res["comm_input"].tobytes()).hexdigest(),
                             "proof_head_hex": res["proof"][:72].hex()})
    return out


def snark_vectors():
    """whole-SNARK and computation-commitment digests of the same instances, from the C oracle"""
    seed_c, seed_p = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
    out = {"_source": "oracle/spark.c + oracle/sat.c (C restatement of my_lib_prove incl. SPARK) on instances from "
                      "tests/gadgets_model.py; seed_commit = bytes(range(64)), seed_proof = bytes((7*i+3)%256)",
           "cases": []}
    cases = [("add6", GM.build_point_add(GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3))),
             ("mult1", GM.build_point_mult(GM.synthetic_mult_ops(0x5650494E + 2, 1, weights=[(1 << 127) + 12345])))]
    for name, g in cases:
        inst = GM.instance_new(g)
        res = O.snark_prove(inst, seed_c, seed_p)
        assert O.snark_verify(inst, res) == 1
        out["cases"].append({"name": name, "snark_len": len(res["proof"]),
                             "snark_sha256": hashlib.sha256(res["proof"]).hexdigest(),
                             "comm_len": len(res["comm"]), "comm_sha256": hashlib.sha256(res["comm"]).hexdigest()})
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "snark_digests.json"), "w") as f:
        json.dump(snark_vectors(), f, indent=1)
    with open(os.path.join(HERE, "sumcheck_vectors.json"), "w") as f:
        json.dump(sumcheck_vectors(), f, indent=1)
    with open(os.path.join(HERE, "sat_proof_digests.json"), "w") as f:
        json.dump(proof_vectors(), f, indent=1)
    print("golden vectors written")
