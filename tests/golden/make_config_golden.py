"""Oracle digests of the BASELINE configurations at their real sizes -> tests/golden/config_digests.json.

    python tests/golden/make_config_golden.py all            # every case below, one process each
    python tests/golden/make_config_golden.py A mult         # one case, merged into the JSON

Chain (no product code between the witness inputs and the digest): synthetic witness inputs (label ->
weights, points) -> tests/gadgets_model.py (Python model of point_mult.rs / point_addition.rs
and of Instance::new) -> oracle/ (C restatement of SNARK::encode + my_lib_prove) -> SHA-256 of the SNARK
bytes, the computation commitment and the two witness commitments.  The GPU tests rebuild the same inputs,
run vpin_gadget_point_*_dev + vpin_snark_prove_dev and compare digests (tests/test_gpu_configs.py).
Op counts: point_mult.rs:27-67, point_addition.rs:38-70, src/LeNet/Server.py:690-698,753-761.
(The reference itself cannot produce vectors here: it is Rust and no toolchain exists)
"""
import hashlib
import json
import os
import resource
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

OUT = os.environ.get("VPIN_GOLDEN_OUT", os.path.join(HERE, "config_digests.json"))
SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))

# (label, kind): every non-empty instance of BASELINE.json's five configs the oracle can prove in this
# container (L5-mult, 2^25 constraints, needs ~170 GB: see `l5` below)
CASES = [("3_32", "add"), ("3_32", "mult"), ("7_256", "add"), ("7_256", "mult"), ("A", "add"), ("A", "mult"),
         ("E", "add"), ("L1", "add"), ("L2", "add"), ("L3", "add"), ("L4", "add"), ("L5", "add"), ("L6", "add"),
         ("L7", "add"), ("L7", "mult"), ("L6", "mult"), ("L1", "mult"), ("E", "mult"), ("L3", "mult")]


def inputs_digest(kind, inp):
    h = hashlib.sha256()
    if kind == "mult":
        w, x, y = inp
        h.update(b"".join(int(v).to_bytes(16, "little") for v in w))
        h.update(x.tobytes())
        h.update(y.tobytes())
    else:
        for a in inp:
            h.update(a.tobytes())
    return h.hexdigest()


def model_instance(kind, inp):
    """witness inputs -> padded instance through the Python model of the reference's gadgets"""
    import gadgets_model as GM

    def ints(a):
        return [int.from_bytes(bytes(r), "little") for r in a]
    if kind == "mult":
        w, x, y = inp
        g = GM.build_point_mult(list(zip([int(v) for v in w], ints(x), ints(y))))
    else:
        px, py, rx, ry, rz = inp
        g = GM.build_point_add(list(zip(ints(px), ints(py), ints(rx), ints(ry), [int(v) for v in rz])))
    return GM.instance_new(g)


def one(label, kind):
    import gadgets_model as GM
    import oracle_lib as O
    from vpin_amd import gadgets as G
    t0 = time.time()
    inp = G.synthetic_mult_inputs(label) if kind == "mult" else G.synthetic_add_inputs(label)
    # the inputs come from the library's generator: pin its first points to the Python model's
    cfg = G.CONFIGS[label]
    if kind == "mult":
        ref = GM.synthetic_points(G.SEED + cfg["index"], 2)
        got = [(int.from_bytes(bytes(inp[1][i]), "little"), int.from_bytes(bytes(inp[2][i]), "little")) for i in range(2)]
        assert ref == got
    inst = model_instance(kind, inp)
    t1 = time.time()
    res = O.snark_prove(inst, SEED_C, SEED_P, threads=os.cpu_count() or 1)
    t2 = time.time()
    assert O.snark_verify(inst, res) == 1
    ent = {"label": label, "kind": kind, "ops": len(inp[0]),
           "num_cons": inst["num_cons"], "num_vars": inst["num_vars"],
           "num_cons_unpadded": inst["num_cons_unpadded"], "nnz": [len(inst[k][0]) for k in "ABC"],
           "inputs_sha256": inputs_digest(kind, inp),
           "snark_len": len(res["proof"]), "snark_sha256": hashlib.sha256(res["proof"]).hexdigest(),
           "comm_len": len(res["comm"]), "comm_sha256": hashlib.sha256(res["comm"]).hexdigest(),
           "comm_para_sha256": hashlib.sha256(res["comm_para"].tobytes()).hexdigest(),
           "comm_input_sha256": hashlib.sha256(res["comm_input"].tobytes()).hexdigest(),
           "snark_head_hex": res["proof"][:64].hex(), "snark_tail_hex": res["proof"][-64:].hex(),
           "oracle_s": {"model_instance": round(t1 - t0, 1), "encode_prove": round(t2 - t1, 1),
                        "maxrss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}}
    merge(f"{label}-{kind}", ent)
    print(json.dumps(ent))


def l5_comm():
    """L5-mult (6000 ops, 2^25 constraints): the oracle's prover does not fit this container, its
    SNARK::encode does.  The committed digest of the computation commitment pins vpin_spark_encode_dev at
    full size; the GPU test then feeds the HIP proof + this commitment to the oracle's verifier."""
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    from vpin_amd import gadgets as G
    t0 = time.time()
    inp = G.synthetic_mult_inputs("L5")
    inst = model_instance("mult", inp)
    t1 = time.time()
    L = O.lib()
    O._spark_decl(L)
    r = O.make_r1cs(inst)
    ccap = L.oracle_spark_comm_bytes(C.byref(r))
    comm = np.zeros(ccap, dtype=np.uint8)
    clen = C.c_size_t(0)
    dec = L.oracle_spark_encode(C.byref(r), os.cpu_count() or 1, comm.ctypes.data_as(C.c_void_p), ccap, C.byref(clen))
    assert dec
    L.oracle_spark_decomm_free(dec)
    cb = bytes(comm[:clen.value])
    ent = {"label": "L5", "kind": "mult", "ops": len(inp[0]), "num_cons": inst["num_cons"], "num_vars": inst["num_vars"],
           "num_cons_unpadded": inst["num_cons_unpadded"], "nnz": [len(inst[k][0]) for k in "ABC"],
           "inputs_sha256": inputs_digest("mult", inp), "comm_len": len(cb), "comm_sha256": hashlib.sha256(cb).hexdigest(),
           "oracle_s": {"model_instance": round(t1 - t0, 1), "encode": round(time.time() - t1, 1),
                        "maxrss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}}
    merge("L5-mult", ent)
    print(json.dumps(ent))


def l5_sat():
    """L5-mult: the oracle's R1CS satisfiability proof (my_R1CSProof_prove, commit_test.rs:136-334) fits this container
    where its whole SNARK does not.  A SNARK's first bytes ARE that proof (lib.rs:330-338: r1cs_sat_proof is the first field),
    so its digest pins the sat half of the HIP SNARK -- both ZK sum-checks at 2^25 constraints, the witness commitments,
    the evaluation proof -- byte for byte."""
    import oracle_lib as O
    from vpin_amd import gadgets as G
    t0 = time.time()
    inp = G.synthetic_mult_inputs("L5")
    inst = model_instance("mult", inp)
    t1 = time.time()
    res = O.sat_prove(inst, SEED_C, SEED_P, threads=os.cpu_count() or 1)
    assert O.sat_verify(inst, res) == 1
    with open(OUT) as f:
        doc = json.load(f)
    ent = doc["cases"]["L5-mult"]
    assert ent["inputs_sha256"] == inputs_digest("mult", inp)
    ent.update({"sat_len": len(res["proof"]), "sat_sha256": hashlib.sha256(res["proof"]).hexdigest(),
                "comm_para_sha256": hashlib.sha256(res["comm_para"].tobytes()).hexdigest(),
                "comm_input_sha256": hashlib.sha256(res["comm_input"].tobytes()).hexdigest(),
                "inst_evals_sha256": hashlib.sha256(res["inst_evals"].tobytes()).hexdigest(),
                "oracle_sat_s": {"model_instance": round(t1 - t0, 1), "sat_prove": round(time.time() - t1, 1),
                                 "maxrss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}})
    merge("L5-mult", ent)
    print(json.dumps(ent))


def l5_full():
    """L5-mult, the whole SNARK: ~95 GB and ~25 core-minutes in the oracle, more than the build container has, so this
    mode is run on the GPU box's HOST cores (322 GB, 16 cores; nothing touches the GPU):
        gpurun --timeout 1200 -- 'VPIN_GOLDEN_OUT=gpurun_out/l5full.json python tests/golden/make_config_golden.py l5full'
    and the entry it prints is merged into config_digests.json here.  Same chain as every other case (Python gadget model ->
    oracle); it re-derives the commitment and sat-half digests the container produced and refuses to write if they differ."""
    import threading
    import oracle_lib as O
    from vpin_amd import gadgets as G
    stop = threading.Event()

    def beat():
        t = time.time()
        while not stop.wait(60):
            print(f"[l5full] {time.time() - t:.0f} s, maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB", flush=True)
    threading.Thread(target=beat, daemon=True).start()
    t0 = time.time()
    inp = G.synthetic_mult_inputs("L5")
    inst = model_instance("mult", inp)
    t1 = time.time()
    print(f"[l5full] instance built in {t1 - t0:.0f} s", flush=True)
    res = O.snark_prove(inst, SEED_C, SEED_P, threads=int(os.environ.get("VPIN_ORACLE_THREADS", os.cpu_count() or 1)))
    t2 = time.time()
    print(f"[l5full] proved in {t2 - t1:.0f} s", flush=True)
    ok = O.snark_verify(inst, res)
    stop.set()
    assert ok == 1
    with open(os.path.join(HERE, "config_digests.json")) as f:
        ent = json.load(f)["cases"]["L5-mult"]
    assert ent["inputs_sha256"] == inputs_digest("mult", inp)
    assert ent["comm_sha256"] == hashlib.sha256(res["comm"]).hexdigest(), "computation commitment differs from the container's"
    assert ent["sat_sha256"] == hashlib.sha256(res["proof"][:ent["sat_len"]]).hexdigest(), "sat half differs from the container's"
    ent.update({"snark_len": len(res["proof"]), "snark_sha256": hashlib.sha256(res["proof"]).hexdigest(),
                "snark_head_hex": res["proof"][:64].hex(), "snark_tail_hex": res["proof"][-64:].hex(),
                "oracle_full_s": {"where": "GPU box host cores (EPYC 9575F share)", "model_instance": round(t1 - t0, 1),
                                  "encode_prove": round(t2 - t1, 1), "verify": round(time.time() - t2, 1),
                                  "maxrss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}})
    merge("L5-mult", ent)
    print(json.dumps(ent))


def merge(key, ent):
    doc = {"_source": "oracle/ (C restatement of SNARK::encode + my_lib_prove) on instances built by tests/gadgets_model.py "
                      "from the synthetic witness inputs of vpin_amd/gadgets.py CONFIGS; seed_commit = bytes(range(64)), "
                      "seed_proof = bytes((7*i+3)%256); regenerate with tests/golden/make_config_golden.py",
           "cases": {}}
    if os.path.exists(OUT):
        with open(OUT) as f:
            doc = json.load(f)
    doc["cases"][key] = ent
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "all":
        for lab, kind in CASES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), lab, kind])
    elif sys.argv[1] == "l5":
        l5_comm()
    elif sys.argv[1] == "l5sat":
        l5_sat()
    elif sys.argv[1] == "l5full":
        l5_full()
    else:
        one(sys.argv[1], sys.argv[2])
