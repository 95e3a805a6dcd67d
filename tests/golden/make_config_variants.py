"""More than one transcript per configuration -> tests/golden/config_variants.json.

    python tests/golden/make_config_variants.py            # every case below (about ten minutes of the oracle on 8 cores)

tests/golden/config_digests.json pins every instance of every BASELINE configuration under ONE seed pair and on
satisfied witnesses.  This file adds, through the same chain (synthetic inputs -> tests/gadgets_model.py -> oracle/):

  seeds     conv f=3 and CNN A (both gadgets) under two more (seed_commit, seed_proof) pairs: other blinds, other
            challenges, other round polynomials from the first transcript byte on;
  unsat     witnesses that do NOT satisfy their instance, at configuration size -- the reference's prover does not check
            (is_sat is an assert of the gadget builders, point_mult.rs:650-651), a proof comes out, and the leading-
            coefficient shortcut of the device's phase-1 rounds must step aside:
              * built from inputs (device path): a point addition with R == P and a point multiplication of a point with
                y == 0 -- the gadgets' inverse-of-zero convention (Scalar::invert(0) = 0) breaks `c * (Rx - Px) = 1`
                resp. `c * 2Py = 1`;
              * an assignment entry changed after synthesis (host-buffer path), conv f=3 and CNN A's 2^20-constraint instance;
  rz_heavy  conv f=7's point additions with 86 of 96 accumulators at infinity (rz = 1 fraction 86/96), the
            `t2 = px * rz` / `(1 - rz)` branches of the gadget carrying the result.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_config_golden as MG  # noqa: E402

OUT = os.path.join(HERE, "config_variants.json")

SEED_PAIRS = {
    "s1": (bytes((13 * i + 1) % 256 for i in range(64)), bytes((31 * i + 7) % 256 for i in range(64))),
    "s2": (hashlib.sha512(b"vpin seed_commit 2").digest(), hashlib.sha512(b"vpin seed_proof 2").digest()),
}


def variant_inputs(name):
    """-> (kind, witness inputs, host-side tamper or None): the ONE definition the tests rebuild their inputs from"""
    from vpin_amd import gadgets as G
    if name in ("3_32-add#RequalsP", ):
        px, py, rx, ry, rz = G.synthetic_add_inputs("3_32")
        rx[5], ry[5], rz[5] = px[5], py[5], 0          # R == P: the chord's denominator vanishes
        return "add", (px, py, rx, ry, rz), None
    if name == "3_32-mult#yzero":
        w, x, y = G.synthetic_mult_inputs("3_32")
        y[2] = 0                                        # the doubling's denominator 2Py vanishes in the first step
        w = list(w)
        w[2] = 5
        return "mult", (w, x, y), None
    if name == "7_256-add#rz86":
        px, py, rx, ry, rz = G.synthetic_add_inputs("7_256")
        idx = [i for i in range(96) if i % 10 != 9][:86]
        for i in idx:
            rz[i] = 1
            rx[i] = 0
            ry[i] = 0
        assert int(rz.sum()) == 86
        return "add", (px, py, rx, ry, rz), None
    label, rest = name.split("-", 1)
    kind, tag = rest.split("#")
    inp = G.synthetic_mult_inputs(label) if kind == "mult" else G.synthetic_add_inputs(label)
    if tag == "tampered":
        # one entry of the assignment changed after synthesis: position = a third of the way into the unpadded variables
        return kind, inp, "third"
    return kind, inp, None


def tamper(inst):
    """vars_input[k] += 1 and vars[k] += 1 (vars = para + input stays consistent), k = num_vars_unpadded // 3"""
    import pymodel as M
    k = inst["num_vars_unpadded"] // 3
    for key in ("vars_input", "vars"):
        tab = np.array(inst[key], dtype=np.uint64).reshape(-1, 4)
        v = (M.from_mont_limbs(tab[k]) + 1) % M.Q
        tab[k] = M.to_mont_limbs(v)
        inst[key] = tab
    return k


CASES = [(f"{lab}-{kind}#{s}", s) for lab in ("3_32", "A") for kind in ("add", "mult") for s in SEED_PAIRS] + [
    ("3_32-add#RequalsP", None), ("3_32-mult#yzero", None), ("3_32-mult#tampered", None), ("3_32-add#tampered", None),
    ("A-mult#tampered", None), ("7_256-add#rz86", None),
    # round 5: the large instances under a second seed pair too -- CNN E's and LeNet layer 3's point-mult
    # instances (2^22 constraints, whole SNARK), and the sat half of LeNet layer 5's (2^25; `l5sat` below)
    ("E-mult#s1", "s1"), ("L3-mult#s1", "s1")]


def one(name, seed_key):
    import oracle_lib as O
    t0 = time.time()
    kind, inp, tam = variant_inputs(name)
    inst = MG.model_instance(kind, inp)
    tam_at = tamper(inst) if tam else None
    sc, sp = SEED_PAIRS[seed_key] if seed_key else (MG.SEED_C, MG.SEED_P)
    t1 = time.time()
    res = O.snark_prove(inst, sc, sp, threads=os.cpu_count() or 1)
    sat = O.is_sat(inst) if hasattr(O, "is_sat") else None
    ok = O.snark_verify(inst, res)
    ent = {"kind": kind, "ops": len(inp[0]), "num_cons": inst["num_cons"], "num_vars": inst["num_vars"],
           "seed_commit_hex": sc.hex(), "seed_proof_hex": sp.hex(), "inputs_sha256": MG.inputs_digest(kind, inp),
           "tampered_at": tam_at, "oracle_is_sat": sat, "oracle_verifier_accepts": int(ok),
           "snark_len": len(res["proof"]), "snark_sha256": hashlib.sha256(res["proof"]).hexdigest(),
           "comm_sha256": hashlib.sha256(res["comm"]).hexdigest(),
           "comm_para_sha256": hashlib.sha256(res["comm_para"].tobytes()).hexdigest(),
           "comm_input_sha256": hashlib.sha256(res["comm_input"].tobytes()).hexdigest(),
           "oracle_s": round(time.time() - t1, 1), "model_s": round(t1 - t0, 1)}
    print(name, json.dumps(ent), flush=True)
    return ent


def l5_sat_variant(seed_key="s1"):
    """L5-mult (6000 ops, 2^25 constraints) under a second seed pair: the oracle's R1CS satisfiability proof (it fits this
    container, the whole SNARK does not) = the first sat_len bytes of the SNARK (lib.rs:330-338) -> key L5-mult#sat_<seed>"""
    import oracle_lib as O
    from vpin_amd import gadgets as G
    t0 = time.time()
    inp = G.synthetic_mult_inputs("L5")
    inst = MG.model_instance("mult", inp)
    sc, sp = SEED_PAIRS[seed_key]
    t1 = time.time()
    res = O.sat_prove(inst, sc, sp, threads=os.cpu_count() or 1)
    assert O.sat_verify(inst, res) == 1
    ent = {"kind": "mult", "ops": len(inp[0]), "num_cons": inst["num_cons"], "num_vars": inst["num_vars"],
           "seed_commit_hex": sc.hex(), "seed_proof_hex": sp.hex(), "inputs_sha256": MG.inputs_digest("mult", inp),
           "sat_len": len(res["proof"]), "sat_sha256": hashlib.sha256(res["proof"]).hexdigest(),
           "comm_para_sha256": hashlib.sha256(res["comm_para"].tobytes()).hexdigest(),
           "comm_input_sha256": hashlib.sha256(res["comm_input"].tobytes()).hexdigest(),
           "inst_evals_sha256": hashlib.sha256(res["inst_evals"].tobytes()).hexdigest(),
           "oracle_s": round(time.time() - t1, 1), "model_s": round(t1 - t0, 1)}
    print("L5-mult#sat_" + seed_key, json.dumps(ent), flush=True)
    return ent


def l5_is_sat(seed_key="s1"):
    """oracle_is_sat of the L5-mult#<seed> entry (the 15-minute whole-SNARK run on the GPU box recorded the verifier's
    verdict and left is_sat null).  is_sat does not depend on the seed pair: the model's instance (same inputs digest as the
    entry) through the oracle's is_sat, ~25 GB and a few minutes of this container:
        python tests/golden/make_config_variants.py L5-mult#is_sat"""
    import oracle_lib as O
    from vpin_amd import gadgets as G
    t0 = time.time()
    inp = G.synthetic_mult_inputs("L5")
    inst = MG.model_instance("mult", inp)
    with open(OUT) as f:
        doc = json.load(f)
    ent = doc["cases"]["L5-mult#" + seed_key]
    assert ent["inputs_sha256"] == MG.inputs_digest("mult", inp) and ent["num_cons"] == inst["num_cons"]
    t1 = time.time()
    ent["oracle_is_sat"] = int(O.is_sat(inst))
    ent["is_sat_s"] = round(time.time() - t1, 1)
    print("L5-mult#" + seed_key, "oracle_is_sat", ent["oracle_is_sat"], f"(model {t1 - t0:.0f} s, is_sat {ent['is_sat_s']} s)", flush=True)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    return ent


def l5_full_variant(seed_key="s1"):
    """L5-mult's WHOLE SNARK under a second seed pair -> key L5-mult#<seed>.  ~95 GB and ~12 minutes of the oracle: run on the GPU
    box's host cores (nothing touches the GPU), the entry goes to $VPIN_GOLDEN_OUT and is merged into config_variants.json here:
        gpurun --timeout 1200 -- 'VPIN_GOLDEN_OUT=gpurun_out/l5_s1.json python tests/golden/make_config_variants.py L5-mult#s1'
    It re-derives the sat half the container produced (L5-mult#sat_<seed>) and refuses to write if that differs."""
    import resource
    import threading
    import oracle_lib as O
    from vpin_amd import gadgets as G
    stop = threading.Event()

    def beat():
        t = time.time()
        while not stop.wait(60):
            print(f"[L5-mult#{seed_key}] {time.time() - t:.0f} s, maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB", flush=True)
    threading.Thread(target=beat, daemon=True).start()
    t0 = time.time()
    inp = G.synthetic_mult_inputs("L5")
    inst = MG.model_instance("mult", inp)
    sc, sp = SEED_PAIRS[seed_key]
    t1 = time.time()
    print(f"[L5-mult#{seed_key}] instance built in {t1 - t0:.0f} s", flush=True)
    res = O.snark_prove(inst, sc, sp, threads=int(os.environ.get("VPIN_ORACLE_THREADS", os.cpu_count() or 1)))
    t2 = time.time()
    ok = O.snark_verify(inst, res)
    stop.set()
    assert ok == 1
    with open(OUT) as f:
        sat = json.load(f)["cases"]["L5-mult#sat_" + seed_key]
    assert sat["inputs_sha256"] == MG.inputs_digest("mult", inp)
    assert sat["sat_sha256"] == hashlib.sha256(res["proof"][:sat["sat_len"]]).hexdigest(), "sat half differs from the container's"
    ent = {"kind": "mult", "ops": len(inp[0]), "num_cons": inst["num_cons"], "num_vars": inst["num_vars"],
           "seed_commit_hex": sc.hex(), "seed_proof_hex": sp.hex(), "inputs_sha256": MG.inputs_digest("mult", inp),
           "tampered_at": None, "oracle_is_sat": None, "oracle_verifier_accepts": int(ok),
           "snark_len": len(res["proof"]), "snark_sha256": hashlib.sha256(res["proof"]).hexdigest(),
           "comm_sha256": hashlib.sha256(res["comm"]).hexdigest(),
           "comm_para_sha256": hashlib.sha256(res["comm_para"].tobytes()).hexdigest(),
           "comm_input_sha256": hashlib.sha256(res["comm_input"].tobytes()).hexdigest(),
           "oracle_s": round(t2 - t1, 1), "model_s": round(t1 - t0, 1),
           "where": "GPU box host cores", "maxrss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}
    print("L5-mult#" + seed_key, json.dumps(ent), flush=True)
    out = os.environ.get("VPIN_GOLDEN_OUT")
    if out:
        with open(out, "w") as f:
            json.dump({"L5-mult#" + seed_key: ent}, f, indent=1, sort_keys=True)
    return ent


def main():
    if sys.argv[1:] == ["L5-mult#is_sat"]:
        l5_is_sat("s1")
        sys.exit(0)
    if sys.argv[1:] == ["L5-mult#s1"]:   # the GPU box's host cores only: see l5_full_variant
        l5_full_variant("s1")
        return
    doc = {"_source": "tests/golden/make_config_variants.py: oracle/ on instances built by tests/gadgets_model.py; see its docstring",
           "cases": {}}
    only = sys.argv[1:]
    if only and os.path.exists(OUT):
        doc = json.load(open(OUT))
    for name, s in CASES:
        if only and name not in only:
            continue
        doc["cases"][name] = one(name, s)
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    if not only or "L5-mult#sat_s1" in only:
        doc["cases"]["L5-mult#sat_s1"] = l5_sat_variant("s1")
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
