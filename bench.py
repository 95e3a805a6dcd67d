#!/usr/bin/env python3
"""bench.py -- Spartan sat-proof throughput of the MI355X hot path on vPIN's CNN-A trace shape.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  Prints ONE JSON line on rank 0.

A "step" is one pass of the hot path over one batch of synthetic input: the two R1CS
satisfiability proofs vPIN produces for one trace (vPIN_proof_generation/src/main.rs:14-46):
the point-multiplication instance and the point-addition instance of the configured network
(default: CNN network A = BASELINE.json configs[1]: 178 point-mults -> 616,592 constraints
(2^20 padded) and 2,144 point-adds -> 21,440 constraints (2^15 padded)).  The witness is
synthetic (seeded SplitMix64 points on the curve E2, gadget-generated tables -- SURVEY.md 8(d)).
value = unpadded R1CS constraints proven per second, whole job (all ranks).  Multi-GPU: the
independent (trace) instances shard across ranks with no data-path collective -> weak scaling.

Beside the headline value the line carries
  roofline     : the fused sum-check round kernel (sc_cubic_fused), algorithmic bytes / HIP-event
                 time over the timed region, against the 8 TB/s HBM3E peak;
  cpu_baseline : the CPU oracle (a C restatement of the reference prover, oracle/) timed on
                 this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--label", default="A", help="trace label: 3_32, A, 7_256, E, L5")
    ap.add_argument("--n-mult", type=int, default=0, help="override the number of point multiplications")
    ap.add_argument("--n-add", type=int, default=0, help="override the number of point additions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mult", type=int, default=18, help="point-mults in the CPU baseline sample")
    ap.add_argument("--cpu-sample-add", type=int, default=256)
    ap.add_argument("--serial", action="store_true", help="prove the two instances one after the other")
    ap.add_argument("--host-buffers", action="store_true",
                    help="time vpin_sat_prove (instance + witness start in host memory: PCIe-inclusive; never the headline)")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the roofline kernel from a separate rocprofv3 --pmc pass")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import vpin_amd
    from vpin_amd import gadgets as G

    from vpin_amd.dist import Group, env_rank

    rank, local_rank, world = env_rank()
    if world > 1:
        torch.cuda.set_device(local_rank)
    grp = Group(backend="nccl", device=torch.device("cuda", local_rank) if world > 1 else None)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # ---- synthetic workload (host side, outside the timed region) ----
    t0 = time.perf_counter()
    mult = G.synthetic_mult_instance(args.label, args.n_mult or None)
    add = G.synthetic_add_instance(args.label, args.n_add or None)
    insts = [mult.as_dict(), add.as_dict()]
    cons = [mult.num_cons_unpadded, add.num_cons_unpadded]
    setup_s = time.perf_counter() - t0

    # one context (stream, pool, generator tables) per instance: the two proofs of a trace are
    # independent, so they run concurrently on two host threads and share the GPU
    ctxs = [vpin_amd.Context(local_rank) for _ in insts]
    ctx = ctxs[0]

    def barrier():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    # instance + the three assignments resident in HBM before the timed region (the PCIe-inclusive
    # variant is vpin_sat_prove; its rate is noted in DESIGN.md)
    resident = []
    for cx, d in zip(ctxs, insts):
        resident.append((cx.r1cs_upload(d), cx.upload(d["vars_para"]), cx.upload(d["vars_input"]),
                         cx.upload(d["vars"]), d["inputs"]))

    last_spans = {}
    names = ("mult", "add")

    def prove_one(k):
        cx, d, (di, tp, ti, tv, inp) = ctxs[k], insts[k], resident[k]
        if args.host_buffers:
            r = cx.sat_prove(d, SEED_C, SEED_P)
        else:
            r = cx.sat_prove_resident(di, tp, ti, tv, inp, SEED_C, SEED_P)
        last_spans[names[k]] = cx.sat_timings()  # thread-local in the library: read on the proving thread
        return r

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=len(insts)) if not args.serial else None

    def step():
        if pool is None:
            return [prove_one(k) for k in range(len(insts))]
        return list(pool.map(prove_one, range(len(insts))))  # ctypes releases the GIL during the calls

    for _ in range(args.warmup):
        proofs = step()
    if args.warmup == 0:
        proofs = None

    for cx in ctxs:
        cx.prof_reset()
        cx.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proofs = step()
    for cx in ctxs:
        cx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    stats = {}
    for cx in ctxs:
        for name, v in cx.prof_read().items():
            a = stats.setdefault(name, {"launches": 0, "ms": 0.0, "alg_bytes": 0.0})
            for kk in a:
                a[kk] += v[kk]
        cx.prof_enable(False)

    elapsed = grp.max_over_ranks(elapsed)

    total_cons = sum(cons) * args.steps * world
    value = total_cons / elapsed

    line = {
        "metric": "R1CS constraints/sec, Spartan sat proof (vPIN point-mult + point-add instances)",
        "value": value,
        "unit": "constraints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u256 (mod q = 2^252+..., mod p = 2^255-19; 32-bit limbs)",
        "data": "synthetic",
        "config": {
            "workload": f"vPIN trace '{args.label}': sat proof of the point-mult instance + the point-add instance",
            "label": args.label,
            "point_mults": cons[0] // 3464, "point_adds": cons[1] // 10,
            "constraints_unpadded": cons, "constraints_padded": [insts[0]["num_cons"], insts[1]["num_cons"]],
            "scope": "R1CSProof (commitments + both ZK sum-checks + evaluation proof); SPARK encode/eval proof not included",
            "parallelism": f"instances sharded over {world} rank(s), no collective; per rank the two proofs of the trace run "
                           + ("serially" if args.serial else "concurrently (2 host threads, 2 HIP streams)"),
        },
    }

    # ---- roofline of the fused sum-check round kernel ----
    k = stats.get("sc_cubic_fused")
    if k and k["ms"] > 0:
        achieved = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9
        line["roofline"] = {
            "kernel": "sc_bind_eval_kernel<4> (fused fold + cubic round evaluation, phase 1)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": args.pmc_traffic,
            "launches": k["launches"], "avg_launch_us": k["ms"] * 1e3 / k["launches"],
            "alg_bytes_per_launch": k["alg_bytes"] / k["launches"],
        }
    line["kernels"] = {name: {"launches": v["launches"], "ms": round(v["ms"], 4),
                              "GBps_alg": (v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] else None}
                       for name, v in stats.items()}
    line["spans_ms_last_step"] = {n: {kk: round(vv * 1e3, 3) for kk, vv in sp.items()} for n, sp in last_spans.items()}
    line["setup_s"] = round(setup_s, 3)
    line["inputs"] = "host buffers (PCIe-inclusive, CSR/CSC built per proof)" if args.host_buffers else "resident in HBM"
    # HBM traffic of the roofline kernel from the committed rocprofv3 --pmc passes (separate runs)
    if "roofline" in line and line["roofline"]["traffic"] is None:
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                line["roofline"]["traffic"] = json.load(f).get("sc_bind_eval_kernel<4>", {}).get("hbm_bytes_per_launch")

    # ---- CPU baseline: the oracle on a bounded sample, rank 0, N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        threads = min(os.cpu_count() or 1, 16)
        sm = G.synthetic_mult_instance(args.label, min(args.cpu_sample_mult, cons[0] // 3464))
        sa = G.synthetic_add_instance(args.label, min(args.cpu_sample_add, cons[1] // 10))
        sample_cons = sm.num_cons_unpadded + sa.num_cons_unpadded
        t0 = time.perf_counter()
        r1 = O.sat_prove(sm.as_dict(), SEED_C, SEED_P, threads=threads)
        tm = O.sat_timings()
        r2 = O.sat_prove(sa.as_dict(), SEED_C, SEED_P, threads=threads)
        cpu_s = time.perf_counter() - t0
        assert len(r1["proof"]) and len(r2["proof"])
        line["cpu_baseline"] = {
            "value": sample_cons / cpu_s, "unit": "constraints/s", "cores": threads, "kind": "port",
            "sample": f"first {sm.num_cons_unpadded // 3464} point-mults + first {sa.num_cons_unpadded // 10} point-adds of "
                      f"the same trace ({sample_cons} constraints), C oracle, OpenMP rows in the commitment "
                      f"(as rayon in the reference), single-threaded sum-checks; {cpu_s:.1f} s",
            "spans_ms_mult": {kk: round(vv * 1e3, 1) for kk, vv in tm.items()},
        }
        sm.free()
        sa.free()

    if rank == 0:
        print(json.dumps(line))
    if pool is not None:
        pool.shutdown()
    for cx in ctxs:
        cx.close()
    grp.close()


if __name__ == "__main__":
    main()
