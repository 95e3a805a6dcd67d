"""One SNARK of one instance with the library's trace switches on (development aid):
python tools/trace_one.py <label> <mult|add> [n_ops]   (VPIN_SPARK_TRACE=1 VPIN_TAIL_TRACE=1 in the environment)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpin_amd  # noqa: E402
from vpin_amd import gadgets as G  # noqa: E402

lab, kind = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else None
SEED_C, SEED_P = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
with vpin_amd.Context(0) as ctx:
    inp = G.synthetic_mult_inputs(lab, n) if kind == "mult" else G.synthetic_add_inputs(lab, n)
    g = ctx.gadget_point_mult_dev(*inp) if kind == "mult" else ctx.gadget_point_add_dev(*inp)
    dec, comm = g.spark_encode()
    for it in range(3):
        t0 = time.perf_counter()
        r = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
        print(f"iteration {it}: {1e3 * (time.perf_counter() - t0):.2f} ms, {len(r['proof'])} bytes", ctx.spark_timings(), file=sys.stderr)
    dec.free()
    g.free()
