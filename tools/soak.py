"""Repeat one SNARK many times and compare the bytes (development aid: a visibility race in a mailbox or a finisher would
show up as a rare difference): python tools/soak.py <label> <mult|add> <iterations>"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpin_amd  # noqa: E402
from vpin_amd import gadgets as G  # noqa: E402

lab, kind, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
SEED_C, SEED_P = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
with vpin_amd.Context(0) as ctx:
    inp = G.synthetic_mult_inputs(lab) if kind == "mult" else G.synthetic_add_inputs(lab)
    g = ctx.gadget_point_mult_dev(*inp) if kind == "mult" else ctx.gadget_point_add_dev(*inp)
    dec, comm = g.spark_encode()
    want = None
    bad = 0
    for it in range(iters):
        r = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
        h = hashlib.sha256(r["proof"]).hexdigest()
        if want is None:
            want = h
        elif h != want:
            bad += 1
            print(f"iteration {it}: {h} != {want}", flush=True)
    print(f"{lab}-{kind}: {iters} proofs, {bad} differing, sha256 {want}")
    dec.free()
    g.free()
sys.exit(1 if bad else 0)
