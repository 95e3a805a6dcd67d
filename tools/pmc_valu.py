"""VALU-issue occupancy of the long kernels from ONE rocprofv3 --pmc pass (SQ counters) -> profiles/r03_pmc_valu.json.

usage: pmc_valu.py <pmc_dir> [bench_json_of_the_same_command]
Counters (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count QUAD-cycles summed over waves,
GRBM_GUI_ACTIVE is summed over the 8 XCDs):
  SQ_INSTS_VALU        vector ALU wave-instructions issued
  SQ_ACTIVE_INST_VALU  quad-cycles in which a wave had a VALU instruction executing
  SQ_WAVE_CYCLES       quad-cycles of wave lifetime;  SQ_WAIT_INST_ANY / SQ_WAIT_ANY: issue stalls / parked waves
  GRBM_GUI_ACTIVE      busy cycles, x8 XCDs
Derived per kernel class (sums over its dispatches):
  kernel_cycles        = GRBM_GUI_ACTIVE / 8
  valu_issue_frac      = 4 * SQ_ACTIVE_INST_VALU / (kernel_cycles * CUs * 4 SIMDs): share of SIMD-cycles with a VALU instruction
                         in flight -- the measured counterpart of bench.py's static "VALU instructions per table addition x 4
                         cycles" ceiling
  cycles_per_valu_inst = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU  (4.0 = every instruction a plain 4-cycle one; the 64-bit
                         multiply-adds take longer)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", os.environ.get("VPIN_PMC_VALU_OUT", "r03_pmc_valu.json"))
CUS = 256


def short(name):
    return name.replace("vpin::", "").replace("void ", "").split("(")[0]


def main():
    d = sys.argv[1]
    bench = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    assert f, f"no counter_collection.csv under {d}"
    per = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    res = {}
    for k, c in per.items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0 or c.get("SQ_INSTS_VALU", 0.0) <= 0:
            continue
        ent = {"dispatches": len(ndisp[k]), "kernel_cycles": cyc, "SQ_INSTS_VALU": c["SQ_INSTS_VALU"],
               "SQ_ACTIVE_INST_VALU_quadcycles": c.get("SQ_ACTIVE_INST_VALU"),
               "SQ_WAVE_CYCLES_quadcycles": c.get("SQ_WAVE_CYCLES"),
               "valu_issue_frac": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (cyc * CUS * 4),
               "cycles_per_valu_inst": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_INSTS_VALU"],
               "valu_insts_per_simd_cycle": c["SQ_INSTS_VALU"] / (cyc * CUS * 4)}
        if c.get("SQ_WAVE_CYCLES"):
            ent["wave_cycle_shares"] = {n: c.get(n, 0.0) / c["SQ_WAVE_CYCLES"] for n in
                                        ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if n in c}
        res[k] = ent
    if bench:
        proofs = int(bench.get("steps", 1)) + int(bench.get("warmup", 1))  # the profiled command proves the instance this often
        for s in bench.get("roofline", {}).get("secondary", []):
            if s["kernel"].startswith("msm_rows_kernel") and s.get("table_adds"):
                hot = sum(v["SQ_INSTS_VALU"] for k, v in res.items() if k.startswith("msm_rows_hot"))
                allk = sum(v["SQ_INSTS_VALU"] for k, v in res.items() if k.startswith("msm_rows"))
                # wave-instructions x 64 lanes / lane-level table additions of ONE proof
                res["_msm_rows_valu_instructions_per_table_add"] = {
                    "all_row_commitments": allk / proofs * 64.0 / s["table_adds"],
                    "static_isa_count_of_the_window_loop": s.get("valu_instructions_per_add"),
                    "proofs_in_the_profiled_command": proofs,
                    "note": "SQ_INSTS_VALU of every msm_rows* dispatch / proofs x 64 lanes / the table additions bench.py counted for "
                            "one proof's row commitments (vpin_prof_enable level 2).  The derefs commitment (msm_rows_hot_kernel: "
                            "full-width scalars, ~22 additions each) sits at the static count; the witness commitments (msm_rows_kernel: "
                            "bits and small values, one or two additions per scalar) pay the per-scalar work -- Montgomery conversion, "
                            "the 22-window digit loop -- for almost no additions, which is what lifts the average"}
                if hot:
                    res["_msm_rows_valu_instructions_per_table_add"]["msm_rows_hot_kernel_wave_instructions_per_proof"] = hot / proofs
    top = dict(sorted(((k, v) for k, v in res.items() if not k.startswith("_")), key=lambda kv: -kv[1]["kernel_cycles"])[:16])
    top.update({k: v for k, v in res.items() if k.startswith("sc_cubic3_kernel")})  # the roofline kernel, however short (bench.py's limiter_frac fallback)
    top.update({k: v for k, v in res.items() if k.startswith("_")})
    doc = {"_how": __doc__.strip().split("\n\n")[0] + "  Command: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES "
                   "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --trace L5 --only mult --serial",
           "kernels": top}
    json.dump(doc, open(OUT, "w"), indent=1)
    print(json.dumps({k: v for k, v in top.items() if k.startswith(("msm_rows", "_msm", "prod_round", "sc_cubic3"))}, indent=1))


if __name__ == "__main__":
    main()
