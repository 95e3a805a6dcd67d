#!/usr/bin/env python3
"""tools/stall_probe.py -- what the card reports while the reference-span pass stalls (round 6).

The serial / four-lane reference-span passes of bench.py showed bimodal times (L5-mult 365 ms or ~950 ms; a lanes pass 590 ms or
1200-1700 ms) with ALL lanes stalled at once for 0.25-1 s, no allocation in the gap and identical work per pass.  This probe proves
the L5-mult instance from witness inputs again and again on one context (gadget -> is_sat -> SNARK::encode -> prove: the span's
first and largest item) while a thread samples every hwmon reading of the card (clocks, power, temperatures) every 2 ms, and
prints the passes with the samples inside the slow ones."""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_common import SEED_C, SEED_P  # noqa: E402


def main():
    import ctypes as C
    import vpin_amd
    from vpin_amd import gadgets as G
    n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    label = sys.argv[2] if len(sys.argv) > 2 else "L5"
    hip = C.CDLL("libamdhip64.so")
    buf = C.create_string_buffer(64)
    assert hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0
    base = "/sys/bus/pci/devices/" + buf.value.decode().lower()
    files = sorted(glob.glob(base + "/hwmon/hwmon*/freq*_input") + glob.glob(base + "/hwmon/hwmon*/power1_*") +
                   glob.glob(base + "/hwmon/hwmon*/temp*_input"))
    files = [f for f in files if not f.endswith(("_cap", "_cap_max", "_cap_min", "_cap_default", "_label"))]
    names = [os.path.basename(f) for f in files]
    print("sampling", names, flush=True)
    samples, stop = [], [False]

    def loop():
        while not stop[0]:
            row = [time.perf_counter()]
            for f in files:
                try:
                    with open(f) as fh:
                        row.append(int(fh.read().strip()))
                except (OSError, ValueError):
                    row.append(-1)
            samples.append(row)
            time.sleep(0.002)
    inp = G.synthetic_mult_inputs(label)
    with vpin_amd.Context(0) as cx:
        g = cx.gadget_point_mult_dev(*inp)
        cx.sat_prepare(g.num_vars)
        g.snark_prove(SEED_C, SEED_P)   # tables, views, pool
        g.free()
        cx.set_expected_proofs(1)
        th = threading.Thread(target=loop, daemon=True)
        th.start()
        passes = []
        ev_files = glob.glob(f"/sys/class/kfd/kfd/proc/{os.getpid()}/stats_*/evicted_ms")

        def evicted():
            tot = 0
            for f in ev_files:
                try:
                    with open(f) as fh:
                        tot += int(fh.read().strip())
                except (OSError, ValueError):
                    pass
            return tot
        print("kfd evicted_ms files:", ev_files, flush=True)
        for k in range(n_pass):
            ev0 = evicted()
            t0 = time.perf_counter()
            g = cx.gadget_point_mult_dev(*inp)
            t1 = time.perf_counter()
            assert g.is_sat()
            t2 = time.perf_counter()
            g.snark_prove(SEED_C, SEED_P)
            t3 = time.perf_counter()
            tm = cx.spark_timings()
            g.free()
            passes.append((t0, t3, (t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, tm["encode"] * 1e3, tm["total"] * 1e3, evicted() - ev0))
            if len(sys.argv) > 3:
                time.sleep(float(sys.argv[3]))   # idle gap between passes (s)
        stop[0] = True
        th.join()
    med = sorted(p[2] for p in passes)[len(passes) // 2]
    print(f"{len(passes)} passes, median {med:.1f} ms; per pass: total | gadget | is_sat | encode | prove")
    for k, p in enumerate(passes):
        slow = p[2] > 1.3 * med
        print(f"pass {k:2d}: {p[2]:8.1f} | {p[3]:6.1f} | {p[4]:5.1f} | {p[5]:7.1f} | {p[6]:7.1f} | kfd evicted +{p[7]} ms" + ("   <-- slow" if slow else ""))
        if (slow or k == 1) and not os.environ.get("PROBE_BRIEF"):
            rows = [r for r in samples if p[0] <= r[0] <= p[1]]
            step = max(1, len(rows) // 40)
            for r in rows[::step]:
                print("      +%7.1f ms  " % ((r[0] - p[0]) * 1e3) + "  ".join(f"{n}={v}" for n, v in zip(names, r[1:])))


if __name__ == "__main__":
    main()
