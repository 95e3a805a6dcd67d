import os, sys, time
sys.path.insert(0, os.getcwd())
import vpin_amd
from vpin_amd import gadgets as G
SEED_C = bytes(range(64)); SEED_P = bytes((7*i+3) % 256 for i in range(64))
lab = sys.argv[1]
with vpin_amd.Context(0) as ctx:
    g = ctx.gadget_point_mult_dev(*G.synthetic_mult_inputs(lab))
    decomm, comm = g.spark_encode()
    for i in range(3):
        if i == 2: os.environ["VPIN_SPARK_TRACE"] = "1"
        t = time.perf_counter()
        r = ctx.snark_prove_resident(g.r1cs, decomm, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
        print("prove ms", (time.perf_counter()-t)*1e3, ctx.sat_timings(), file=sys.stderr)
