"""Shared pieces of bench.py and its other modes (tools/bench_strong.py, tools/bench_concurrent.py): the seeds, the synthetic
trace, the golden digests, the PMC child passes and the power sampler.  Not product code."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))


def _strong_work(args):
    from vpin_amd import gadgets as G
    trace = args.label or args.trace
    labels = list(G.LENET) if trace == "lenet" else trace.split(",")
    work = []
    for lab in labels:
        m = G.synthetic_mult_inputs(lab)
        if m is not None:
            work.append((f"{lab}-mult", "mult", m, 3464 * len(m[0])))
        a = G.synthetic_add_inputs(lab)
        work.append((f"{lab}-add", "add", a, 10 * len(a[4])))
    if args.only:
        work = [w for w in work if w[1] == args.only]
    work.sort(key=lambda w: -w[3])
    return trace, work


def _build_resident(cx, w):
    g = cx.gadget_point_mult_dev(*w[2]) if w[1] == "mult" else cx.gadget_point_add_dev(*w[2])
    cx.sat_prepare(g.num_vars)
    dec, _ = g.spark_encode()
    return g, dec


def _prove_res(cx, g, dec):
    return cx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)


def _golden_digests():
    try:
        with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
            return json.load(f)["cases"]
    except OSError:
        return {}


def live_pmc_traffic(timeout_s=240, cus=256):
    """HBM bytes per launch AND the VALU-issue occupancy of the roofline kernel, MEASURED: three child processes under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE -- the two do not fit one pass -- then the SQ counters; no tracing option
    beside --pmc) prove the largest instance of the trace alone -- the launches `roofline` is scoped to -- and the kernel's
    dispatches are averaged.  gfx950 corrections as in tools/pmc_summary.py / MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB,
    FETCH_SIZE counts half the bytes of wide coalesced reads: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; SQ_ACTIVE_INST_VALU
    counts quad-cycles summed over waves, GRBM_GUI_ACTIVE is summed over the 8 XCDs: valu_issue_frac = 4 * SQ_ACTIVE_INST_VALU /
    (GRBM_GUI_ACTIVE / 8 * CUs * 4 SIMDs).  The caller must have released the GPU's memory (the children build their own tables
    and instance) and must not need the GPU afterwards.  Returns (bytes per launch, dict) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="vpin_pmc_", dir="/tmp")
    kern = "sc_cubic3_kernel<true, true>"
    try:
        for tag, ctrs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                          ("VALU", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])):
            d = os.path.join(tmp, tag)
            cmd = [prof, "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, BENCH,
                   "--trace", "L5", "--only", "mult", "--serial", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-span",
                   "--no-verify", "--no-roofline-pass", "--no-live-pmc", "--sat-only", "--detail-out", os.path.join(d, "child.json")]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                if tag == "VALU" and "FETCH_SIZE" in out and "WRITE_SIZE" in out:
                    out["VALU"] = {"error": f"exit {r.returncode}: {r.stderr[-200:]}"}
                    break
                return None, f"rocprofv3 --pmc {tag}: exit {r.returncode}: {r.stderr[-300:]}"
            files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
            if not files:
                return None, f"rocprofv3 --pmc {tag}: no counter_collection.csv"
            vals = {c: [] for c in ctrs}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] in vals and kern in row["Kernel_Name"]:
                        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            if not all(vals.values()):
                return None, f"rocprofv3 --pmc {tag}: the roofline kernel was not dispatched"
            if tag == "VALU":
                insts, act, gui = (sum(vals[c]) for c in ctrs)
                out["VALU"] = {"dispatches": len(vals[ctrs[0]]), "SQ_INSTS_VALU": insts, "SQ_ACTIVE_INST_VALU_quadcycles": act,
                               "kernel_cycles": gui / 8.0, "valu_issue_frac": 4.0 * act / (gui / 8.0 * cus * 4),
                               "cycles_per_valu_inst": 4.0 * act / insts}
            else:
                out[tag] = {"dispatches": len(vals[tag]), "avg_KiB": sum(vals[tag]) / len(vals[tag])}
        traffic = (2.0 * out["FETCH_SIZE"]["avg_KiB"] + out["WRITE_SIZE"]["avg_KiB"]) * 1024.0
        return traffic, out
    except (subprocess.TimeoutExpired, OSError) as e:
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class PowerSampler:
    """sclk and socket power of THIS rank's card from its hwmon files (sysfs), sampled by a thread while the timed region runs:
    the four-lane step is power-bound, and the line should say at which clock and power it actually ran."""

    def __init__(self, device_index, interval_s=0.05):
        import glob
        self.freq, self.power, self.samples, self.stop_flag, self.th = [], [], [], False, None
        self.interval_s, self.times = interval_s, []
        try:
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            buf = C.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, device_index) == 0:
                base = "/sys/bus/pci/devices/" + buf.value.decode().lower()
                self.freq = glob.glob(base + "/hwmon/hwmon*/freq1_input")
                self.power = glob.glob(base + "/hwmon/hwmon*/power1_average") or glob.glob(base + "/hwmon/hwmon*/power1_input")
        except OSError:
            pass

    @staticmethod
    def _read(paths):
        for p in paths:
            try:
                with open(p) as f:
                    return float(f.read().strip())
            except (OSError, ValueError):
                continue
        return None

    def start(self):
        if not (self.freq or self.power):
            return

        def loop():
            while not self.stop_flag:
                self.times.append(time.perf_counter())
                self.samples.append((self._read(self.freq), self._read(self.power)))
                time.sleep(self.interval_s)
        self.th = threading.Thread(target=loop, daemon=True)
        self.th.start()

    def stop(self, t_from=None, t_to=None):
        """median / extremes over the whole run, or over the samples taken in [t_from, t_to] (time.perf_counter values)"""
        self.stop_flag = True
        if self.th:
            self.th.join()
        pick = [x for t, x in zip(self.times, self.samples) if (t_from is None or t >= t_from) and (t_to is None or t <= t_to)]
        f = sorted(x[0] / 1e6 for x in pick if x[0])
        w = sorted(x[1] / 1e6 for x in pick if x[1])
        if not f and not w:
            return None
        med = lambda v: v[len(v) // 2] if v else None
        return {"samples": len(pick), "sclk_mhz_median": med(f), "sclk_mhz_min": f[0] if f else None, "sclk_mhz_max": f[-1] if f else None,
                "watts_median": med(w), "watts_max": w[-1] if w else None,
                "source": f"hwmon freq1_input / power1_average of this rank's card, every {self.interval_s * 1e3:.0f} ms"}


# ---- the ONE line bench.py prints -----------------------------------------------------------------------------------------
# Round 5's line had grown to 24 KB of nested prose and the driver could not parse it (BENCH_r05.json: parsed = null).  The
# printed line is now a flat, strict-JSON record of at most LINE_MAX bytes; everything else goes to a side file whose path the
# line carries in `detail`.  tests/test_bench_line.py builds a line from a stub and checks both properties.
LINE_MAX = 4096


def _num(x, nd=None):
    """a JSON-safe number (NaN / inf -> None), optionally rounded"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (int, float)):
        if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
            return None
        return round(x, nd) if (nd is not None and isinstance(x, float)) else x
    return None


def _all_true(d):
    return (all(bool(v) for v in d.values()) if isinstance(d, dict) and d else None)


def compact_line(full, detail_path=None):
    """The flat record the driver reads, made from bench.py's full record.  Scalars only inside `roofline` and `cpu_baseline`;
    strings bounded; None for what a run did not measure."""
    rf, cb, cfg = full.get("roofline") or {}, full.get("cpu_baseline") or {}, full.get("config") or {}
    rs = full.get("reference_span") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["metric"] = str(out["metric"])[:160]
    out["config"] = {"workload": str(cfg.get("workload", ""))[:120],
                     "constraints_unpadded_per_step": cfg.get("constraints_unpadded_per_step"),
                     "instances_per_step": len(cfg.get("instances") or {}) or None,
                     "inputs": str(cfg.get("inputs", ""))[:60],
                     "parallelism": str(cfg.get("parallelism_short") or cfg.get("parallelism", ""))[:100],
                     "table_slot_bytes": cfg.get("table_slot_bytes")}
    if rf:
        src = str(rf.get("traffic_source") or "")
        out["roofline"] = {
            "kernel": str(rf.get("kernel", ""))[:80], "bound": rf.get("bound"), "achieved": _num(rf.get("achieved"), 2),
            "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": _num(rf.get("frac"), 4), "traffic": _num(rf.get("traffic"), 0),
            "traffic_how": ("live_pmc" if src.startswith("measured in this run") else "replayed" if src.startswith("replayed")
                            else "caller" if src else None),
            "launches": rf.get("launches"), "avg_launch_us": _num(rf.get("avg_launch_us"), 2),
            "alg_bytes_per_launch": _num(rf.get("alg_bytes_per_launch"), 0),
            "traffic_over_algorithmic": _num(rf.get("traffic_over_algorithmic"), 4),
            "limiter": str(rf.get("limiter", ""))[:24], "limiter_frac": _num(rf.get("limiter_frac"), 4),
            "limiter_how": ("live_pmc" if str(rf.get("limiter_frac_source", "")).startswith("measured in this run") else
                            "replayed" if "replayed" in str(rf.get("limiter_frac_source", "")) else "static_model"),
            "frac_actual": _num(rf.get("frac_actual"), 4),
            "largest_instance_alone_ms": _num(rf.get("largest_instance_alone_ms"), 2),
            "msm_G_adds_s": _num(rf.get("msm_G_adds_s"), 3),
            "msm_frac_of_static_valu_peak": _num(rf.get("msm_frac_of_static_valu_peak"), 4),
            "msm_frac_of_chain": _num(rf.get("msm_frac_of_chain"), 4),
            "msm_steady_G_adds_s": _num(rf.get("msm_steady_G_adds_s"), 3), "msm_watts": _num(rf.get("msm_watts"), 0),
            "msm_sclk_mhz": _num(rf.get("msm_sclk_mhz"), 0),
            "prod_round_frac": _num(rf.get("prod_round_frac"), 4), "prod_round_valu_frac": _num(rf.get("prod_round_valu_frac"), 4),
        }
    if cb:
        out["cpu_baseline"] = {"value": _num(cb.get("value"), 1), "unit": cb.get("unit"), "cores": cb.get("cores"),
                               "kind": cb.get("kind"), "sample": str(cb.get("sample_short") or cb.get("sample", ""))[:120],
                               "seconds": _num(cb.get("seconds"), 2)}
    if rs:
        out["value_reference_span"] = _num(full.get("value_reference_span"), 1)
        out["reference_span_ms"] = _num(rs.get("ms_per_trace"), 1)
        out["reference_span_lanes_ms"] = _num((rs.get("lanes") or {}).get("ms_per_trace"), 1)
        out["reference_span_largest_ms"] = _num(max((rs.get("ms") or {"": None}).values(), key=lambda v: v or 0.0), 1)
        dw = rs.get("dead_work") or {}
        out["reference_span_with_dead_work_ms"] = _num(dw.get("ms_per_trace_with"), 1)
        out["dead_work_how"] = dw.get("digest_how")
        if full.get("span_warning"):
            out["span_warning"] = str(full["span_warning"])[:160]
    pw = full.get("power_during_timed_region") or {}
    out["watts_median"] = _num(pw.get("watts_median"), 0)
    out["sclk_mhz_median"] = _num(pw.get("sclk_mhz_median"), 0)
    if pw.get("watts_median") and full.get("ms_per_step"):
        out["joules_per_step"] = _num(pw["watts_median"] * full["ms_per_step"] * 1e-3, 1)
    hc = full.get("host_cpu") or {}
    if hc:
        out["host_quota_cpus"] = hc.get("quota_cpus")
        out["host_throttled_ms"] = hc.get("throttled_ms_in_timed_region")
    out["hbm_in_use_gib"] = full.get("hbm_in_use_gib_after_timed_region")
    out["window_tables_gib"] = (full.get("hbm_breakdown") or {}).get("window_tables_gib")
    out["bytes_ok"] = _all_true(full.get("bytes_equal_oracle_digest"))
    out["verified_ok"] = _all_true(full.get("verified"))
    enc = full.get("encode_ms") or {}
    if enc:
        out["encode_ms_total"] = _num(sum(enc.values()), 1)
        out["encode_ms_largest"] = _num(max(enc.values()), 1)
    out["proof_bytes_per_step"] = sum((full.get("proof_bytes") or {}).values()) or None
    for k, v in full.items():
        if k.startswith("strong_") and not isinstance(v, (dict, list)):
            out[k] = (str(v)[:160] if isinstance(v, str) else _num(v, 3) if isinstance(v, float) else v)
    if full.get("errors"):
        out["errors"] = "; ".join(f"{k}: {v}" for k, v in full["errors"].items())[:300]
    out["run_s"] = _num(full.get("run_s"), 1)
    out["detail"] = detail_path
    return out


def dumps_line(rec):
    """strict JSON, compact separators; raises when the line is over LINE_MAX bytes or not strictly parseable"""
    s = json.dumps(rec, allow_nan=False, separators=(",", ":"))
    if len(s.encode()) > LINE_MAX:
        raise ValueError(f"bench line is {len(s.encode())} bytes (> {LINE_MAX})")
    if "\n" in s:
        raise ValueError("bench line spans lines")
    json.loads(s, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    return s


def write_detail(full, path):
    """the full record (everything round 5 printed) as a side file; returns the path written, or None"""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1, default=str)
            f.write("\n")
        return path
    except OSError:
        return None


def emit(full, detail_path):
    """write the side file, print the one line; a line that would not fit is cut down to the contract's core instead of failing
    the run (the numbers were measured: they must come out)"""
    import math
    def clean(o):
        if isinstance(o, float):
            return o if math.isfinite(o) else None
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        return o
    full = clean(full)
    written = write_detail(full, detail_path) if detail_path else None
    rel = os.path.relpath(written, ROOT) if written else None
    rec = compact_line(full, rel)
    try:
        s = dumps_line(rec)
    except ValueError as e:
        core = {k: rec.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail")}
        core["line_error"] = str(e)[:100]
        s = json.dumps(core, allow_nan=False, separators=(",", ":"))
    print(s, flush=True)
    return s


def host_cpu_throttle():
    """CFS bandwidth throttling of this process's cgroup so far: {"quota_cpus", "nr_throttled", "throttled_ms"} (cgroup v2
    cpu.stat / cpu.max or v1 cpu/cpu.stat / cpu.cfs_quota_us), or {} where the files are not there.  A one-GPU box gives a job
    16 cores of a 128-thread host: lanes x OpenMP teams + spinning waiters beyond that are stopped for the rest of every 100 ms
    period, which shows as ~10-50 ms stalls of whichever thread happens to run (round 6: asked of the W = 8 rehearsal's stalls)."""
    out = {}
    try:
        for stat, quota in (("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu.max"),
                            ("/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us")):
            if not os.path.exists(stat):
                continue
            kv = {}
            with open(stat) as f:
                for ln in f:
                    p = ln.split()
                    if len(p) == 2:
                        kv[p[0]] = int(p[1])
            out["nr_throttled"] = kv.get("nr_throttled", 0)
            out["throttled_ms"] = kv["throttled_usec"] / 1e3 if "throttled_usec" in kv else kv.get("throttled_time", 0) / 1e6
            try:
                with open(quota) as f:
                    q = f.read().split()
                if quota.endswith("cpu.max"):
                    out["quota_cpus"] = None if q[0] == "max" else int(q[0]) / int(q[1])
                else:
                    per = 100000
                    try:
                        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                            per = int(f2.read())
                    except (OSError, ValueError):
                        pass
                    out["quota_cpus"] = None if int(q[0]) < 0 else int(q[0]) / per
            except (OSError, ValueError, IndexError):
                out["quota_cpus"] = None
            break
        out["affinity_cpus"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    except (OSError, ValueError, KeyError):
        pass
    return out


def thread_cpu_seconds(top=16):
    """[(thread name, tid, user + system CPU seconds)] of this process's threads, busiest first (/proc/self/task/*/stat)"""
    out = []
    tck = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/stat") as f:
                    st = f.read()
                name = st[st.index("(") + 1:st.rindex(")")]
                rest = st[st.rindex(")") + 2:].split()
                out.append((name, int(tid), round((int(rest[11]) + int(rest[12])) / tck, 2)))
            except (OSError, ValueError, IndexError):
                continue
    except OSError:
        return []
    out.sort(key=lambda x: -x[2])
    return out[:top]
