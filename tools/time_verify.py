"""Verification time per instance of a trace (vpin_snark_verify, steady state):
python tools/time_verify.py [trace]        VPIN_VERIFY_TRACE=1 prints the spans of every verification"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpin_amd  # noqa: E402
from vpin_amd import gadgets as G  # noqa: E402

trace = sys.argv[1] if len(sys.argv) > 1 else "lenet"
only = sys.argv[2] if len(sys.argv) > 2 else None
labels = list(G.LENET) if trace == "lenet" else [trace]
SEED_C, SEED_P = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
total = 0.0
with vpin_amd.Context(0) as ctx:
    for lab in labels:
        for kind in ("mult", "add"):
            inp = G.synthetic_mult_inputs(lab) if kind == "mult" else G.synthetic_add_inputs(lab)
            if inp is None or (only and f"{lab}-{kind}" != only):
                continue
            g = ctx.gadget_point_mult_dev(*inp) if kind == "mult" else ctx.gadget_point_add_dev(*inp)
            dec, comm = g.spark_encode()
            r = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
            meta = {"inputs": g.inputs, "num_inputs": g.num_inputs}
            ts = []
            for it in range(3):
                t0 = time.perf_counter()
                ok = ctx.snark_verify(meta, dict(r, comm=comm))
                ts.append(time.perf_counter() - t0)
                if os.environ.get("VPIN_VERIFY_TRACE"):
                    print(f"-- {lab}-{kind} pass {it}: {ts[-1] * 1e3:.2f} ms", file=sys.stderr)
            print(f"{lab}-{kind}: ok={bool(ok)} first {ts[0] * 1e3:.1f} ms, steady {min(ts[1:]) * 1e3:.1f} ms")
            total += min(ts[1:])
            dec.free()
            g.free()
print(f"trace {trace}: {total * 1e3:.1f} ms steady")
