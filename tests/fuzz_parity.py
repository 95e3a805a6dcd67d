"""Differential run beyond the suite's fixed cases: random small instances of both gadgets -- random operation counts, weights
(incl. 0, 1, 2^128 - 1, small and full-width), points (incl. repeated points, R = P, R = -P, y = 0, bytes >= q), rz patterns and
seed pairs -- built on the device, proven as whole SNARKs, and byte-compared with the CPU oracle on instances built by the
Python model of the reference's gadgets (tests/gadgets_model.py, itself pinned to the Rust text by tests/golden/gadget_pins.json).
    python tests/fuzz_parity.py <cases> [seed]
Test infrastructure (it lives under tests/ because it loads the oracle, which only tests/, smoke() and bench.py's cpu_baseline leg
may): the oracle is the CHECKER here."""
import hashlib
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file's own directory)
import vpin_amd  # noqa: E402
import gadgets_model as GM  # noqa: E402
import oracle_lib as O  # noqa: E402

Q = GM.Q


def b32(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def rand_point(rng, pool):
    r = rng.random()
    if r < 0.70:
        return rng.choice(pool)
    if r < 0.80:
        return (rng.choice(pool)[0], 0)                      # y = 0: the doubling's denominator vanishes
    if r < 0.88:
        return (rng.randrange(2**256), rng.randrange(2**256))  # off-curve, bytes >= q: from_bytes_mod_order reduces
    if r < 0.94:
        return (0, 0)
    return (Q - 1, Q - 2)


# ---- --mid: instances of 20..700 point-mults (2^17..2^22 constraints) ---------------------------------------------------------
# The small cases above never leave 2^15 constraints: the hot-column entries of the derefs commitment, the row-per-lane
# (strip) kernel, the column-chunked row kernels, the mid-size round kernels are fuzzed by nothing.  Here the
# oracle side (Python gadget model + C oracle, minutes per case) runs in worker processes on the box's host cores while the
# GPU proves the same inputs; only digests travel.

def mid_ops(seed, it):
    rng = random.Random(seed * 100003 + it)
    pool = GM.synthetic_points(seed + it, 16)
    n = int(round(20 * (35.0 ** rng.random())))          # log-uniform in [20, 700]
    ops = []
    odd = rng.random() < 0.5        # half of the instances keep every point on the curve (satisfied), the others get a few odd ones
    for _ in range(n):
        w = rng.choice([0, 1, 3, 2**128 - 1, rng.randrange(2**16), rng.randrange(2**128), rng.randrange(2**128), rng.randrange(2**128)])
        x, y = rand_point(rng, pool) if (odd and rng.random() < 0.02) else rng.choice(pool)
        ops.append((w, x, y))
    sc, sp = bytes(rng.randrange(256) for _ in range(64)), bytes(rng.randrange(256) for _ in range(64))
    return n, ops, sc, sp


def mid_oracle(args):
    seed, it, threads = args
    n, ops, sc, sp = mid_ops(seed, it)
    t0 = time.time()
    inst = GM.instance_new(GM.build_point_mult([(o[0], o[1] % 2**256, o[2] % 2**256) for o in ops]))
    t1 = time.time()
    exp = O.snark_prove(inst, sc, sp, threads=threads)
    return {"it": it, "n": n, "num_cons": inst["num_cons"], "is_sat": bool(O.is_sat(inst)),
            "proof": hashlib.sha256(exp["proof"]).hexdigest(), "comm": hashlib.sha256(exp["comm"]).hexdigest(),
            "comm_para": hashlib.sha256(exp["comm_para"].tobytes()).hexdigest(),
            "comm_input": hashlib.sha256(exp["comm_input"].tobytes()).hexdigest(),
            "model_s": round(t1 - t0, 1), "oracle_s": round(time.time() - t1, 1)}


def main_mid_oracle(cases, seed, out_path):
    """CPU only: the oracle's digests of the --mid cases -> a JSON fixture (run in the build container; minutes per case)"""
    import json
    threads = int(os.environ.get("VPIN_FUZZ_THREADS", str(os.cpu_count() or 8)))
    doc = {"_source": f"tests/fuzz_parity.py --mid-oracle {cases} {seed}: Python gadget model + C oracle on mid_ops(seed, it)", "seed": seed, "cases": []}
    if os.path.exists(out_path):
        doc = json.load(open(out_path))
    for it in range(len(doc["cases"]), cases):
        doc["cases"].append(mid_oracle((seed, it, threads)))
        with open(out_path, "w") as f:
            json.dump(doc, f, indent=1)
        print(doc["cases"][-1], flush=True)


def main_mid_check(fixture):
    """GPU: prove the fixture's cases (rebuilt from the same seeds) and compare with the committed oracle digests"""
    import json
    doc = json.load(open(fixture))
    seed, bad, taken0, t0 = doc["seed"], 0, 0, time.time()
    with vpin_amd.Context(0) as ctx:
        for exp in doc["cases"]:
            it = exp["it"]
            n, ops, sc, sp = mid_ops(seed, it)
            assert n == exp["n"]
            strip = it % 2 == 1
            if strip:
                os.environ["VPIN_MSM_STRIP_MIN"] = "64"
            try:
                g = ctx.gadget_point_mult_dev([o[0] for o in ops], b32(o[1] for o in ops), b32(o[2] for o in ops))
                try:
                    got = g.snark_prove(sc, sp)
                    sat_dev = g.is_sat()
                finally:
                    g.free()
            finally:
                os.environ.pop("VPIN_MSM_STRIP_MIN", None)
            taken = ctx.strip_rows_taken()
            same = (hashlib.sha256(got["proof"]).hexdigest() == exp["proof"] and hashlib.sha256(got["comm"]).hexdigest() == exp["comm"]
                    and hashlib.sha256(got["comm_para"].tobytes()).hexdigest() == exp["comm_para"]
                    and hashlib.sha256(got["comm_input"].tobytes()).hexdigest() == exp["comm_input"] and sat_dev == exp["is_sat"])
            bad += 0 if same else 1
            print(f"mid case {it}: {n} point-mults, 2^{exp['num_cons'].bit_length() - 1} constraints, sat {exp['is_sat']}, row-per-lane rows "
                  f"{taken - taken0}{' (forced from 64 rows)' if strip else ''}: {'ok' if same else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
            taken0 = taken
    print(f"fuzz_parity --mid-check: {len(doc['cases'])} random point-mult instances of 20..700 operations (seed {seed}), {bad} mismatches "
          "against the oracle's committed digests")
    sys.exit(1 if bad else 0)


def main_mid(cases, seed):
    import multiprocessing as mp
    # a one-GPU box gives 16 host cores whatever os.cpu_count() says: 4 workers x 4 oracle threads by default
    workers = int(os.environ.get("VPIN_FUZZ_WORKERS", "4"))
    threads = int(os.environ.get("VPIN_FUZZ_THREADS", "4"))
    t0 = time.time()
    import threading
    stop = threading.Event()

    def beat():
        while not stop.wait(60):
            print(f"[fuzz --mid] {time.time() - t0:.0f} s, waiting for the oracle workers", flush=True)
    threading.Thread(target=beat, daemon=True).start()
    ctxmp = mp.get_context("spawn")   # the parent holds a GPU context: never fork it
    with ctxmp.Pool(workers) as pool:
        pending = [pool.apply_async(mid_oracle, ((seed, it, threads),)) for it in range(cases)]
        bad, taken0 = 0, 0
        with vpin_amd.Context(0) as ctx:
            for it in range(cases):
                n, ops, sc, sp = mid_ops(seed, it)
                strip = it % 2 == 1
                if strip:
                    os.environ["VPIN_MSM_STRIP_MIN"] = "64"   # the row-per-lane kernel from 64 rows on (read per call)
                try:
                    g = ctx.gadget_point_mult_dev([o[0] for o in ops], b32(o[1] for o in ops), b32(o[2] for o in ops))
                    try:
                        got = g.snark_prove(sc, sp)
                        sat_dev = g.is_sat()
                    finally:
                        g.free()
                finally:
                    os.environ.pop("VPIN_MSM_STRIP_MIN", None)
                taken = ctx.strip_rows_taken()
                exp = pending[it].get(timeout=3600)
                same = (hashlib.sha256(got["proof"]).hexdigest() == exp["proof"] and hashlib.sha256(got["comm"]).hexdigest() == exp["comm"]
                        and hashlib.sha256(got["comm_para"].tobytes()).hexdigest() == exp["comm_para"]
                        and hashlib.sha256(got["comm_input"].tobytes()).hexdigest() == exp["comm_input"] and sat_dev == exp["is_sat"])
                if strip and taken == taken0:   # (not an error: rows with many zero / hot entries stay with the row kernel)
                    print(f"case {it}: the row-per-lane kernel was asked for and took no row of this instance", flush=True)
                taken0 = taken
                if not same:
                    bad += 1
                print(f"mid case {it}: {n} point-mults, 2^{exp['num_cons'].bit_length() - 1} constraints, sat {exp['is_sat']}, strip rows so far {taken}, "
                      f"model {exp['model_s']} s oracle {exp['oracle_s']} s: {'ok' if same else 'MISMATCH'}  [{time.time() - t0:.0f} s]", flush=True)
    stop.set()
    print(f"fuzz_parity --mid: {cases} random point-mult instances of 20..700 operations (seed {seed}), {bad} mismatches against the oracle")
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--mid-oracle":
        return main_mid_oracle(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    if len(sys.argv) > 1 and sys.argv[1] == "--mid-check":
        return main_mid_check(sys.argv[2])
    if len(sys.argv) > 1 and sys.argv[1] == "--mid":
        return main_mid(int(sys.argv[2]) if len(sys.argv) > 2 else 16, int(sys.argv[3]) if len(sys.argv) > 3 else 5)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    rng = random.Random(seed)
    pool = GM.synthetic_points(seed, 24)
    bad, t0 = 0, time.time()
    with vpin_amd.Context(0) as ctx:
        for it in range(cases):
            sc, sp = bytes(rng.randrange(256) for _ in range(64)), bytes(rng.randrange(256) for _ in range(64))
            if rng.random() < 0.45:
                n = rng.choice([1, 1, 2, 3, 5])
                ops = []
                for _ in range(n):
                    w = rng.choice([0, 1, 2, 3, 2**128 - 1, 2**127, rng.randrange(2**16), rng.randrange(2**128), rng.randrange(2**128)])
                    x, y = rand_point(rng, pool)
                    ops.append((w, x, y))
                kind = "mult"
                g = ctx.gadget_point_mult_dev([o[0] for o in ops], b32(o[1] for o in ops), b32(o[2] for o in ops))
                inst = GM.instance_new(GM.build_point_mult([(o[0], o[1] % 2**256, o[2] % 2**256) for o in ops]))
            else:
                n = rng.choice([1, 2, 3, 7, 16, 40, 100, 300])
                ops = []
                for _ in range(n):
                    p, r = rand_point(rng, pool), rand_point(rng, pool)
                    t = rng.random()
                    if t < 0.08:
                        r = p                                   # R = P: inverse of zero
                    elif t < 0.12:
                        r = (p[0], (Q - p[1]) % Q)              # R = -P
                    rz = rng.choice([0, 0, 0, 1, 1, 5])
                    ops.append((p[0], p[1], r[0], r[1], rz))
                kind = "add"
                g = ctx.gadget_point_add_dev(b32(o[0] for o in ops), b32(o[1] for o in ops), b32(o[2] for o in ops), b32(o[3] for o in ops),
                                             np.array([o[4] for o in ops], dtype=np.uint8))
                inst = GM.instance_new(GM.build_point_add([(o[0], o[1], o[2], o[3], 0 if o[4] == 0 else 1) for o in ops]))
            try:
                got = g.snark_prove(sc, sp)
                sat_dev = g.is_sat()
            finally:
                g.free()
            exp = O.snark_prove(inst, sc, sp, threads=8)
            same = (got["proof"] == exp["proof"] and got["comm"] == exp["comm"] and np.array_equal(got["comm_para"], exp["comm_para"])
                    and np.array_equal(got["comm_input"], exp["comm_input"]) and sat_dev == bool(O.is_sat(inst)))
            if not same:
                bad += 1
                print(f"case {it}: MISMATCH {kind} n={n} ops={ops[:2]}... seeds {sc.hex()[:16]} {sp.hex()[:16]}", flush=True)
            if it % 10 == 9:
                print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_parity: {cases} random instances (seed {seed}), {bad} mismatches against the oracle; sha of the run "
          f"{hashlib.sha256(str((cases, seed)).encode()).hexdigest()[:12]}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
