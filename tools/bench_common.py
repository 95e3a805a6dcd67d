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
                   "--no-verify", "--no-roofline-pass", "--no-live-pmc"]
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
