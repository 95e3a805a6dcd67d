"""Repeat ONE cooperative proof many times with W ranks as threads sharing the GPU and compare every rank's bytes with the
single-GPU proof (development aid: a race in the exchange layer, the mailboxes or the sharded round bookkeeping would show as
a rare difference or a timeout): python tools/soak_dist.py <label> <mult|add> <world> <iterations>"""
import hashlib
import os
import sys
import threading

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vpin_amd  # noqa: E402
from vpin_amd import Comm, gadgets as G  # noqa: E402

lab, kind, W, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
SEED_C, SEED_P = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
ctxs = [vpin_amd.Context(0) for _ in range(W)]
inp = G.synthetic_mult_inputs(lab) if kind == "mult" else G.synthetic_add_inputs(lab)
g = ctxs[0].gadget_point_mult_dev(*inp) if kind == "mult" else ctxs[0].gadget_point_add_dev(*inp)
dec, _ = g.spark_encode()
want = hashlib.sha256(ctxs[0].snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)["proof"]).hexdigest()
comms = Comm.local(W)
bad, errs = [0] * W, []


def body(r):
    try:
        ctxs[r].set_comm(comms[r])
        for it in range(iters):
            h = hashlib.sha256(ctxs[r].snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)["proof"]).hexdigest()
            if h != want:
                bad[r] += 1
                print(f"rank {r} iteration {it}: {h} != {want}", flush=True)
        ctxs[r].set_comm(None)
    except BaseException as e:  # noqa: BLE001
        errs.append((r, repr(e)))


ts = [threading.Thread(target=body, args=(r,)) for r in range(W)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"{lab}-{kind} over {W} ranks: {iters} cooperative proofs per rank, {sum(bad)} differing, errors {errs}, sha256 {want}, "
      f"collectives per rank {comms[0].stats()['collectives']}")
sys.exit(1 if (sum(bad) or errs) else 0)
