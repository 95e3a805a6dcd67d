"""bench.py --trace T --concurrent K1,K2,..: K independent copies of a small trace at once on one GPU.  Split out of bench.py
in round 5; bench.py imports it on demand."""
import json
import os
import sys
import threading
import time

from bench_common import BENCH, SEED_C, SEED_P, _build_resident, _golden_digests, _prove_res, _strong_work  # noqa: F401


def main_concurrent(args):
    """--concurrent K[,K2,..]: K independent copies of ONE small trace proven at the same time, each on its own context
    (stream + host thread), whole SNARKs, inputs resident.  A small trace (configs 1-3: conv f=3, CNN A, conv f=7) is a latency
    chain of ~800 host<->device round trips that leaves the chip mostly idle; a service hides the chain by proving several
    traces at once.  Reports the single-trace latency and, per K, the sustained constraints/s and the latency of a trace under
    that load.  One GPU."""
    import hashlib
    trace, work = _strong_work(args)
    total_cons = sum(w[3] for w in work)
    ks = sorted({int(x) for x in str(args.concurrent).split(",") if int(x) > 0})
    if len(ks) > 1:
        # One K per PROCESS: memory a context frees is wiped lazily by the driver and slows the next context's allocations
        # (DESIGN.md section 3), and idle contexts' streams share the hardware queues of the busy ones -- a sweep inside one
        # process measured K = 8 at 1.0x of the single-trace rate where a fresh process gives 3x.  This parent never touches
        # the GPU; it starts one child per K, one after the other, and merges their lines.
        import subprocess
        rows, single, oks, envs = [], None, [], {}
        for K in ks:
            cmd = [sys.executable, BENCH, "--trace", trace, "--concurrent", str(K), "--steps", str(args.steps),
                   "--warmup", str(args.warmup)] + (["--only", args.only] if args.only else [])
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            if out.returncode != 0:
                raise SystemExit(f"--concurrent {K}: child failed\n{out.stderr[-2000:]}")
            d = json.loads(out.stdout.strip().splitlines()[-1])
            rows += d["concurrent"]
            single = single or d["single_trace"]
            oks.append(d["bytes_equal_oracle_digest"])
            envs[K] = {"host_threads_per_context": d["host_threads_per_context"], "GPU_MAX_HW_QUEUES": d["GPU_MAX_HW_QUEUES"]}
        best = max(rows, key=lambda r: r["constraints_per_s"])
        d.update({"value": best["constraints_per_s"], "ms_per_step": best["ms_per_trace_under_load"], "single_trace": single, "concurrent": rows,
                  "bytes_equal_oracle_digest": all(oks), "per_K_environment": envs})
        d["config"]["workload"] = f"vPIN trace '{trace}' x K concurrent copies on one GPU (K contexts / streams / host threads; a fresh process per K), best K = {best['K']}"
        print(json.dumps(d))
        return
    import vpin_amd
    K = ks[0]
    gold = _golden_digests()
    order = [w[0] for w in sorted(work, key=lambda w: -w[3])]
    ctxs = [vpin_amd.Context(0) for _ in range(K)]
    if K > 1:
        for cx in ctxs:
            cx.set_shared_device(True)
    built = [{w[0]: _build_resident(cx, w) for w in work} for cx in ctxs]
    ok, rows = {}, []

    def run_trace(k, n):
        for _ in range(n):
            for name in order:
                r = _prove_res(ctxs[k], *built[k][name])
                ok[(k, name)] = hashlib.sha256(r["proof"]).hexdigest() == gold.get(name, {}).get("snark_sha256")  # every proof made

    run_trace(0, args.warmup)
    t0 = time.perf_counter()
    run_trace(0, args.steps)   # the single-trace latency: one context busy, the others idle
    single_s = (time.perf_counter() - t0) / args.steps
    th = [threading.Thread(target=run_trace, args=(k, args.warmup)) for k in range(K)]   # warm every context
    [t.start() for t in th]
    [t.join() for t in th]
    th = [threading.Thread(target=run_trace, args=(k, args.steps)) for k in range(K)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    el = time.perf_counter() - t0
    rows.append({"K": K, "traces_per_s": K * args.steps / el, "constraints_per_s": K * args.steps * total_cons / el,
                 "ms_per_trace_under_load": el / args.steps * 1e3, "x_single_trace_rate": (K * args.steps / el) * single_s})
    best = max(rows, key=lambda r: r["constraints_per_s"])
    print(json.dumps({
        "metric": "R1CS constraints/sec, whole Spartan SNARK (sat proof + SPARK evaluation proof; vPIN point-mult + point-add instances)",
        "value": best["constraints_per_s"], "unit": "constraints/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": best["ms_per_trace_under_load"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u256 (mod q = 2^252+..., mod p = 2^255-19; 32-bit limbs)", "data": "synthetic",
        "config": {"workload": f"vPIN trace '{trace}' x K concurrent copies on one GPU (K contexts / streams / host threads), best K = {best['K']}",
                   "constraints_unpadded_per_trace": total_cons, "inputs": "resident in HBM"},
        "single_trace": {"ms": single_s * 1e3, "constraints_per_s": total_cons / single_s},
        "concurrent": rows,
        "bytes_equal_oracle_digest": all(ok.values()), "host_cores": os.cpu_count(),
        "host_threads_per_context": os.environ.get("VPIN_HOST_THREADS"), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
