#!/usr/bin/env python3
"""tools/gen_docs.py -- the measured numbers of README.md / DESIGN.md, GENERATED from the committed profiles (round 6).

VERDICT r5: "typed numbers drift" -- README and DESIGN quoted a reference span of 731 ms while the file they cited said 1328.
Every figure the two documents quote for the current round now comes out of this script:

  python tools/gen_docs.py            # rewrite profiles/r06_INDEX.json and the blocks between the GENERATED markers
  python tools/gen_docs.py --check    # exit 1 when a document's block or the index is not what the profiles give (the test)

profiles/r06_INDEX.json maps every quoted figure to the file and the key it was read from; tests/test_docs_generated.py resolves
each entry again and compares the documents' blocks with a fresh generation."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = "r06"
BEGIN = "<!-- BEGIN GENERATED {tag}: tools/gen_docs.py from profiles/{tag}_*.json; do not edit by hand -->"
END = "<!-- END GENERATED {tag} -->"


def load(name):
    with open(os.path.join(P, name)) as f:
        return json.load(f)


def dig(doc, key):
    """dotted path; a component in [brackets] is taken literally (keys with dots or spaces), integers index lists"""
    cur = doc
    parts, buf, depth = [], "", 0
    for ch in key:
        if ch == "[":
            depth += 1
            if depth == 1:
                if buf:
                    parts.append(buf)
                buf = ""
                continue
        if ch == "]":
            depth -= 1
            if depth == 0:
                parts.append(buf)
                buf = ""
                continue
        if ch == "." and depth == 0:
            if buf:
                parts.append(buf)
            buf = ""
            continue
        buf += ch
    if buf:
        parts.append(buf)
    for p in parts:
        cur = cur[int(p)] if isinstance(cur, list) else cur[p]
    return cur


class Index:
    def __init__(self):
        self.entries = []
        self.cache = {}

    def get(self, ident, file, key, unit="", nd=None):
        if file not in self.cache:
            self.cache[file] = load(file)
        v = dig(self.cache[file], key)
        self.entries.append({"id": ident, "value": v, "unit": unit, "file": "profiles/" + file, "key": key})
        if nd is not None and isinstance(v, (int, float)):
            return round(v, nd)
        return v


def build():
    ix = Index()
    D = f"{TAG}_bench_default.json"
    g = ix.get
    ms = g("lenet_ms_per_step", D, "ms_per_step", "ms", 1)
    val = g("lenet_constraints_per_s", D, "value", "constraints/s")
    steps, warm = g("lenet_steps", D, "steps"), g("lenet_warmup", D, "warmup")
    run_s = g("bench_run_s", D, "run_s", "s", 1)
    frac = g("roofline_frac", D, "roofline.frac", "", 3)
    ach = g("roofline_achieved_GBps", D, "roofline.achieved", "GB/s", 0)
    traffic = g("roofline_traffic_bytes", D, "roofline.traffic", "B")
    alg = g("roofline_alg_bytes_per_launch", D, "roofline.alg_bytes_per_launch", "B")
    tfrac = g("roofline_traffic_over_algorithmic", D, "roofline.traffic_over_algorithmic", "", 3)
    fact = g("roofline_frac_actual", D, "roofline.frac_actual", "", 3)
    lim = g("roofline_valu_issue_occupancy", D, "roofline.limiter_frac", "", 3)
    avg_us = g("roofline_avg_launch_us", D, "roofline.avg_launch_us", "us", 1)
    launches = g("roofline_launches", D, "roofline.launches")
    msm_g = g("msm_G_adds_s", D, "roofline.msm_G_adds_s", "G/s", 1)
    msm_fr = g("msm_frac_of_static_valu_peak", D, "roofline.msm_frac_of_static_valu_peak", "", 3)
    msm_w = g("msm_steady_watts", D, "roofline.msm_watts", "W", 0)
    msm_clk = g("msm_steady_sclk_mhz", D, "roofline.msm_sclk_mhz", "MHz", 0)
    msm_st = g("msm_steady_G_adds_s", D, "roofline.msm_steady_G_adds_s", "G/s", 1)
    prod_fr = g("prod_round_frac", D, "roofline.prod_round_frac", "", 3)
    alone = g("l5_mult_alone_ms", D, "roofline.largest_instance_alone_ms", "ms", 1)
    span = g("reference_span_ms", D, "reference_span.ms_per_trace", "ms", 1)
    span_v = g("reference_span_constraints_per_s", D, "value_reference_span", "constraints/s")
    span_l5 = g("reference_span_l5_mult_ms", D, "reference_span.ms.[L5-mult]", "ms", 1)
    lanes = g("reference_span_lanes_ms", D, "reference_span.lanes.ms_per_trace", "ms", 1)
    lanes_v = g("reference_span_lanes_constraints_per_s", D, "reference_span.lanes.constraints_per_s", "constraints/s")
    dead = g("reference_span_with_dead_work_ms", D, "reference_span.dead_work.ms_per_trace_with", "ms", 0)
    dead_how = g("dead_work_digest_how", D, "reference_span.dead_work.digest_how")
    third = g("third_commitment_ms_total", D, "reference_span.dead_work.third_commitment_ms_total", "ms", 1)
    enc_l5 = g("encode_ms_l5_mult", D, "encode_ms.[L5-mult]", "ms", 1)
    hbm = g("hbm_in_use_gib", D, "hbm_in_use_gib_after_timed_region", "GiB")
    tabs = g("hbm_window_tables_gib", D, "hbm_breakdown.window_tables_gib", "GiB", 1)
    resid = g("hbm_resident_inputs_gib", D, "hbm_breakdown.resident_inputs_gib", "GiB", 1)
    cached = g("hbm_cached_temporaries_gib", D, "hbm_breakdown.cached_temporaries_gib", "GiB", 1)
    watts = g("step_watts_median", D, "power_during_timed_region.watts_median", "W", 0)
    sclk = g("step_sclk_mhz_median", D, "power_during_timed_region.sclk_mhz_median", "MHz", 0)
    cpu_v = g("cpu_baseline_constraints_per_s", D, "cpu_baseline.value", "constraints/s")
    cpu_c = g("cpu_baseline_cores", D, "cpu_baseline.cores")
    cpu_s = g("cpu_baseline_seconds", D, "cpu_baseline.seconds", "s", 1)
    cpu_n = g("cpu_baseline_sample_constraints", D, "cpu_baseline.small_sample.constraints")
    quota = g("host_cpu_quota", D, "host_cpu.quota_cpus")
    thr = g("host_throttled_ms_in_timed_region", D, "host_cpu.throttled_ms_in_timed_region", "ms", 0)
    joules = round(watts * ms * 1e-3, 0) if watts else None
    cfg = {}
    for lab in ("3_32", "A", "7_256", "E", "L5_mult", "sat_only", "default_no_cu_split"):
        f = f"{TAG}_bench_{lab}.json"
        cfg[lab] = (g(f"{lab}_ms_per_step", f, "ms_per_step", "ms", 2), g(f"{lab}_constraints_per_s", f, "value", "constraints/s"))
    hb = {}
    for lab in ("A_hostbuf_sat", "A_hostbuf_sat_pinned", "A_hostbuf_snark", "E_hostbuf_snark", "L5_hostbuf_snark"):
        hb[lab] = g(f"{lab}_ms_per_step", f"{TAG}_bench_{lab}.json", "ms_per_step", "ms", 2)
    # the rocprofv3 kernel trace of the same command: the roofline kernel on the 2^25 instance's first stream, and the shares
    S = f"{TAG}_rocprofv3_summary.json"
    summ = load(S)
    by = summ["kernel_stats_by_stream"]
    cands = [i for i, e in enumerate(by) if "sc_cubic3_kernel<true, true>" in e["name"]]
    i1 = min(cands, key=lambda i: int(by[i]["stream"]))   # the lowest stream id: the 2^25 instance's first stream
    sid = by[i1]["stream"]
    kt_avg = round(g("ktrace_roofline_kernel_avg_ns_stream1", S, f"kernel_stats_by_stream.{i1}.avg_ns", "ns") / 1e3, 2)
    kt_calls = g("ktrace_roofline_kernel_calls_stream1", S, f"kernel_stats_by_stream.{i1}.calls")

    def share(sub, ident):
        i = next(i for i, e in enumerate(summ["kernel_stats"]) if sub in e["name"])
        return g(ident, S, f"kernel_stats.{i}.pct", "%", 2)
    sh_hot, sh_strip, sh_rows = share("msm_rows_hot_kernel", "share_msm_rows_hot"), share("msm_strip_kernel", "share_msm_strip"), share("msm_rows_kernel<", "share_msm_rows")
    sh_tail, sh_bul, sh_fin = share("spark_tail_kernel", "share_spark_tail"), share("bullet_step_kernel", "share_bullet_step"), share("round_finish_kernel", "share_round_finish")
    sh_rf = share("sc_cubic3_kernel<true, true>", "share_roofline_kernel")
    reh = {}
    for W in (2, 4, 8):
        f = f"{TAG}_strong_rehearse{W}_lenet.json"
        reh[W] = (g(f"rehearse{W}_model_ms", f, "model_ms", "ms", 1), g(f"rehearse{W}_lower_bound_ms", f, "model_ms_lower_bound_quietest_pass", "ms", 1),
                  g(f"rehearse{W}_host_threads", f, "host_cpu.env.VPIN_HOST_THREADS"))

    M = lambda x: f"{x / 1e6:.1f}"
    L = []
    L.append(f"Round 6, one MI355X, `profiles/{TAG}_*` (index of every figure: `profiles/{TAG}_INDEX.json`; this block is written by "
             "`tools/gen_docs.py` and checked by `tests/test_docs_generated.py`).")
    L.append("")
    L.append("| quantity | value | from |")
    L.append("|---|---|---|")
    L.append(f"| LeNet trace, 12 whole SNARKs, inputs resident (`python3 bench.py --gpus 1 --steps {steps} --warmup {warm}`, the driver's command) | "
             f"**{ms} ms/step = {M(val)} M constraints/s** | `{D}`: `ms_per_step`, `value` |")
    others = []
    for tag in ("box1", "box2"):
        f = f"{TAG}_bench_default_{tag}.json"
        others.append((f, g(f"{tag}_ms_per_step", f, "ms_per_step", "ms", 1), g(f"{tag}_constraints_per_s", f, "value", "constraints/s"),
                       g(f"{tag}_roofline_frac", f, "roofline.frac", "", 3)))
    s96 = f"{TAG}_bench_default_slot96.json"
    s96_ms, s96_val = g("slot96_ms_per_step", s96, "ms_per_step", "ms", 1), g("slot96_constraints_per_s", s96, "value", "constraints/s")
    s96_hbm, s96_tab = g("slot96_hbm_in_use_gib", s96, "hbm_in_use_gib_after_timed_region", "GiB"), g("slot96_window_tables_gib", s96, "hbm_breakdown.window_tables_gib", "GiB", 1)
    slotb = g("table_slot_bytes", D, "config.table_slot_bytes", "B")
    L.append(f"| ... window tables in {slotb}-byte slots (bench.py's choice where every rank owns its GPU); the library's default layout, 96 bytes, "
             f"on the same box minutes later (`--table-slot 96`) | {s96_ms} ms/step = {M(s96_val)} M constraints/s with {s96_tab} GiB of tables, {s96_hbm} GiB in use | `{s96}` |")
    L.append("| ... the same command on two other boxes the same day, earlier trees of the round (96-byte slots, same kernels; boxes, and one box "
             "over 20 minutes, differ by 2-4 %: 362.8-378.8 ms/step over the round's 96-byte runs) | "
             + "; ".join(f"{m} ms/step = {M(v)} M constraints/s, roofline frac {fr}" for _, m, v, fr in others)
             + " | " + ", ".join(f"`{f}`" for f, _, _, _ in others) + " |")
    L.append(f"| ... the whole `bench.py` run, printed line {os.path.getsize(os.path.join(P, TAG + '_bench_default.line'))} bytes | {run_s} s | `run_s`; `{TAG}_bench_default.line` |")
    L.append(f"| socket power / shader clock over the timed region (medians) | {watts:.0f} W, {sclk:.0f} MHz ({joules:.0f} J per trace) | `power_during_timed_region` |")
    L.append(f"| roofline kernel `sc_cubic3_kernel<true,true>`: algorithmic bytes / HIP-event time | {ach:.0f} GB/s = **{frac} of 8 TB/s** "
             f"({launches} launches, {avg_us} us each, {alg / 1e6:.1f} MB algorithmic per launch) | `roofline.*` |")
    L.append(f"| ... HBM bytes per launch, PMC, measured by the run itself | {traffic / 1e6:.1f} MB = {tfrac} x algorithmic ({fact} of peak really moved) | `roofline.traffic` |")
    L.append(f"| ... share of SIMD-cycles with a VALU instruction in flight (PMC) | {lim} | `roofline.limiter_frac` |")
    U = f"{TAG}_bench_default_under_rocprof.json"
    u_us, u_n = g("under_rocprof_avg_launch_us", U, "roofline.avg_launch_us", "us", 2), g("under_rocprof_launches", U, "roofline.launches")
    L.append(f"| ... the same kernel in `rocprofv3 --kernel-trace --stats` of `bench.py --steps 3` a minute later on that box (stream {sid}: the "
             f"2^25 instance before the other lanes start, all six proofs of the process) | {kt_avg} us over {kt_calls} launches in the trace; "
             f"that run's own HIP-event figure: {u_us} us over the {u_n} launches of its timed region | `{S}`, `{U}` |")
    L.append(f"| row-commitment MSM, 2^25 instance alone | {msm_g} G table additions/s = {msm_fr} of the static VALU-issue bound; "
             f"3 s loop: {msm_st} G/s at {msm_clk:.0f} MHz, {msm_w:.0f} W | `roofline.msm_*` |")
    L.append(f"| product rounds >= 2^20 pairs | {prod_fr} of 8 TB/s | `roofline.prod_round_frac` |")
    L.append(f"| kernel-time shares of a default step (kernel trace) | MSM {sh_hot} + {sh_strip} + {sh_rows} % (hot rows, strips, rows); `spark_tail` {sh_tail} %, "
             f"`bullet_step` {sh_bul} %, `round_finish` {sh_fin} %; the roofline kernel {sh_rf} % | `{S}` |")
    L.append(f"| reference's own span per instance, serially (witness inputs -> gadget -> is_sat -> `SNARK::encode` -> prove -> bytes) | "
             f"**{span} ms per trace = {M(span_v)} M constraints/s** (L5-mult {span_l5} ms) | `reference_span` |")
    L.append(f"| ... the instances on the bench's four streams | {lanes} ms = {M(lanes_v)} M constraints/s | `reference_span.lanes` |")
    L.append(f"| ... with the reference's unused third commitment ({third} ms, measured) and zlib digest ({dead_how}) put back | {dead / 1e3:.1f} s | `reference_span.dead_work` |")
    L.append(f"| `SNARK::encode` of the 2^25 instance (outside the timed region) | {enc_l5} ms | `encode_ms` |")
    L.append(f"| L5-mult (2^25 constraints) alone, one stream | {alone} ms ({cfg['L5_mult'][0]} ms in `{TAG}_bench_L5_mult.json`) | `roofline.largest_instance_alone_ms` |")
    L.append(f"| HBM in use after a step | {hbm} GiB: window tables {tabs}, resident inputs {resid}, cached temporaries {cached} | `hbm_breakdown` |")
    L.append(f"| CPU baseline: the C oracle (restated reference prover), {cpu_n} constraints, whole SNARKs | {cpu_v / 1e3:.1f} k constraints/s on {cpu_c} threads ({cpu_s} s) | `cpu_baseline` |")
    L.append(f"| host: CFS quota of the box / throttled time inside the timed region | {quota:.0f} CPUs / {thr:.0f} ms | `host_cpu` |")
    L.append("")
    L.append("| trace (whole SNARKs, resident) | ms/step | M constraints/s | file |")
    L.append("|---|---|---|---|")
    for lab, name in (("3_32", "conv f=3, 32x32 (configs[0])"), ("A", "CNN A (configs[1])"), ("7_256", "conv f=7, 256x256 (configs[2])"),
                      ("E", "CNN E (configs[3])"), ("L5_mult", "LeNet L5-mult alone, serial"), ("default_no_cu_split", "LeNet, `--cu-split none`")):
        L.append(f"| {name} | {cfg[lab][0]} | {M(cfg[lab][1])} | `{TAG}_bench_{lab}.json` |")
    L.append(f"| LeNet, sat proofs only (`--sat-only`, SURVEY 8(a) rows H1-H10) | {cfg['sat_only'][0]} | {M(cfg['sat_only'][1])} | `{TAG}_bench_sat_only.json` |")
    L.append("")
    L.append(f"From HOST buffers (PCIe-inclusive, never the headline): CNN A's sat proofs {hb['A_hostbuf_sat']} ms ({hb['A_hostbuf_sat_pinned']} ms page-locked); whole SNARKs with "
             f"`SNARK::encode` per proof: A {hb['A_hostbuf_snark']} ms, E {hb['E_hostbuf_snark']} ms, L5 {hb['L5_hostbuf_snark']} ms (`{TAG}_bench_*_hostbuf_*.json`).")
    L.append("")
    L.append("One LeNet trace over W GPUs -- a MODEL from sections measured on one GPU (`bench.py --scaling strong --rehearse W`), no multi-GPU hardware involved: "
             + "; ".join(f"W = {W}: {reh[W][0]} ms (lower bound {reh[W][1]}, {reh[W][2]} host thread(s) per rank)" for W in (2, 4, 8))
             + f" (`{TAG}_strong_rehearse{{2,4,8}}_lenet.json`).")
    return "\n".join(L), ix.entries


def splice(text, block):
    b, e = BEGIN.format(tag=TAG), END.format(tag=TAG)
    if b not in text or e not in text:
        raise SystemExit(f"markers of {TAG} not found")
    i, j = text.index(b), text.index(e)
    return text[:i] + b + "\n" + block + "\n" + text[j:]


def main():
    check = "--check" in sys.argv
    block, entries = build()
    index = {"_how": "tools/gen_docs.py: every figure README.md / DESIGN.md quote for round 6, with the file and key it is read from",
             "entries": entries}
    bad = []
    ipath = os.path.join(P, f"{TAG}_INDEX.json")
    want = json.dumps(index, indent=1) + "\n"
    if check:
        if not os.path.exists(ipath) or open(ipath).read() != want:
            bad.append(ipath)
    else:
        with open(ipath, "w") as f:
            f.write(want)
    for doc in ("README.md", "DESIGN.md"):
        path = os.path.join(ROOT, doc)
        old = open(path).read()
        new = splice(old, block)
        if new != old:
            if check:
                bad.append(path)
            else:
                with open(path, "w") as f:
                    f.write(new)
    if bad:
        print("out of date:", bad)
        sys.exit(1)
    print(block if not check else "documents and index match the profiles")


if __name__ == "__main__":
    main()
