#!/usr/bin/env python3
"""tools/ab.py -- same-box A/B runs of bench.py (one recipe for every profiles/*_ab_*.txt file).

  python3 tools/ab.py TAG [--repeat N] [--common "ARGS"] -- "name|ENV=1 ENV2=x|--bench --args" ...

Every variant is one child process `python3 bench.py <common> <args>` with <env> added; the variants run in the given
order, the whole list N times (interleaved, so drift of the box hits every variant alike).  Per run one summary line
(ms/step, the largest instance's spans, the digests' verdict) goes to stdout and to gpurun_out/TAG/summary.txt; the
bench lines themselves to gpurun_out/TAG/<name>.<k>.json.  The parent never touches the GPU.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_COMMON = "--steps 10 --warmup 2 --no-cpu-baseline --no-span --no-roofline-pass --no-live-pmc"


def main():
    argv = sys.argv[1:]
    if "--" not in argv or not argv:
        raise SystemExit(__doc__)
    k = argv.index("--")
    head, variants = argv[:k], argv[k + 1:]
    tag, repeat, common = head[0], 1, DEFAULT_COMMON
    i = 1
    while i < len(head):
        if head[i] == "--repeat":
            repeat = int(head[i + 1])
        elif head[i] == "--common":
            common = head[i + 1]
        i += 2
    out_dir = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out_dir, exist_ok=True)
    summ = open(os.path.join(out_dir, "summary.txt"), "a")

    def say(msg):
        print(msg, flush=True)
        summ.write(msg + "\n")
        summ.flush()

    say(f"# {tag}: {time.strftime('%Y-%m-%d %H:%M:%S')}  common: {common}")
    for rep in range(repeat):
        for v in variants:
            name, env_s, args_s = (v.split("|") + ["", ""])[:3]
            env = dict(os.environ)
            for kv in env_s.split():
                kk, vv = kv.split("=", 1)
                env[kk] = vv
            cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + common.split() + args_s.split()
            t0 = time.time()
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            except subprocess.TimeoutExpired:
                say(f"{name:28s} TIMEOUT")
                continue
            with open(os.path.join(out_dir, f"{name}.{rep}.err"), "w") as f:
                f.write(r.stderr)
            if r.returncode != 0:
                say(f"{name:28s} FAILED rc={r.returncode}: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''}")
                continue
            line = r.stdout.strip().splitlines()[-1]
            with open(os.path.join(out_dir, f"{name}.{rep}.json"), "w") as f:
                f.write(line + "\n")
            d = json.loads(line)
            sp = d.get("spans_ms_last_step", {})
            big = max(sp, key=lambda n: sp[n].get("spark_total", sp[n].get("total", 0.0))) if sp else None
            b = sp.get(big, {})
            ok = d.get("bytes_equal_oracle_digest")
            ok_s = "-" if ok is None else ("ok" if (all(ok.values()) if isinstance(ok, dict) else ok) else "DIFF")
            pw = d.get("power_during_timed_region") or {}
            rf = d.get("roofline") or {}
            say(f"{name:28s} {d['ms_per_step']:8.2f} ms/step  {big}: total {b.get('spark_total', b.get('total'))} sat {b.get('spark_sat', b.get('total'))} "
                f"derefs {b.get('spark_derefs_commit')} net {b.get('spark_network_build')} prod {b.get('spark_product_layer')} hash {b.get('spark_hash_layer')}"
                f"  bytes {ok_s}  sclk {pw.get('sclk_mhz_median')} W {pw.get('watts_median')}  roof {rf.get('frac', 0):.3f}"
                f"  hbm {d.get('hbm_in_use_gib_after_timed_region')}  [{time.time() - t0:.0f} s]")


if __name__ == "__main__":
    main()
