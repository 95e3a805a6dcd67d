"""HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/r02_pmc_traffic.json.

usage: pmc_summary.py <section> <fetch_dir> <write_dir> [bench_json_of_the_same_command]
The counters are in KiB.  On gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM
section), so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The correction is calibrated for 16-byte-per-lane
streaming reads; for the MSM's 96-byte random table gathers it is an upper bound (noted in the entry).
The product rounds of >= 2^20 pairs run under their own kernel names (prod_round_kernel<*, true, true>), so their class
is the one bench.py's roofline.secondary times.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", os.environ.get("VPIN_PMC_OUT", "r02_pmc_traffic.json"))


def short(name):
    return name.replace("vpin::", "").replace("void ", "").split("(")[0]


def load(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    assert f, f"no counter_collection.csv under {d}"
    rows = []
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    return rows


def main():
    section, fdir, wdir = sys.argv[1:4]
    bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    # the two passes run the same deterministic command: dispatch k of kernel X in one is dispatch k of X in the other
    per = defaultdict(lambda: {"f": [], "w": [], "grid": []})
    for _, name, grid, v in fetch:
        per[name]["f"].append(v)
        per[name]["grid"].append(grid)
    for _, name, grid, v in write:
        per[name]["w"].append(v)
    res = {}
    for name, d in per.items():
        n = min(len(d["f"]), len(d["w"]))
        if n == 0:
            continue
        if name.startswith("msm_rows_kernel"):
            big = [i for i in range(n) if d["f"][i] * 1024 >= 1e9]  # the row commitments of the large polynomials
            if big:
                bf, bw = sum(d["f"][i] for i in big) / len(big), sum(d["w"][i] for i in big) / len(big)
                res["msm_rows_kernel (>= 1 GB fetched)"] = {
                    "dispatches": len(big), "FETCH_SIZE_avg_KiB": bf, "WRITE_SIZE_avg_KiB": bw,
                    "hbm_bytes_per_launch": (bf + bw) * 1024,
                    "note": "random 96-byte table gathers: FETCH_SIZE taken as is (the x2 correction of MI355X_MICROARCH.md is calibrated "
                            "for 16-byte-per-lane streaming reads)"}
        f, w = sum(d["f"][:n]) / n, sum(d["w"][:n]) / n
        ent = {"dispatches": n, "FETCH_SIZE_avg_KiB": f, "WRITE_SIZE_avg_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
        if name.startswith("msm_"):
            ent["note"] = "96-byte random table gathers: the x2 FETCH_SIZE correction is calibrated for streaming reads only, so this is an upper bound"
        res[name] = ent
    # the >= 2^20-pair launches of the product rounds run under their own kernel names (template parameter BIG)
    bigs = [v for k, v in per.items() if k.startswith("prod_round_kernel<") and k.rstrip().endswith("true>") and k.count("true") + k.count("false") == 3]
    if bigs:
        fs = [x for d in bigs for x in d["f"]]
        ws = [x for d in bigs for x in d["w"]]
        n = min(len(fs), len(ws))
        res["prod_round_kernel<*, true, true> (>= 2^20 pairs)"] = {
            "dispatches": n, "FETCH_SIZE_avg_KiB": sum(fs[:n]) / n, "WRITE_SIZE_avg_KiB": sum(ws[:n]) / n,
            "hbm_bytes_per_launch": (2 * sum(fs[:n]) / n + sum(ws[:n]) / n) * 1024}
    if bench:
        for s in bench.get("roofline", {}).get("secondary", []):
            key = "msm_rows_kernel (>= 1 GB fetched)"
            if s["kernel"].startswith("msm_rows_kernel") and key in res and section != "bench_default":
                # the section's command proves ONE instance per step: its roofline pass counted the additions of the same launches
                res[key]["table_adds_per_launch"] = s["table_adds"] / s["launches"]
                res[key]["bytes_per_table_add"] = res[key]["hbm_bytes_per_launch"] / res[key]["table_adds_per_launch"]
                res[key]["table_adds_note"] = ("vpin_prof_enable level 2 count in the same command's roofline pass, over its "
                                               f"{s['launches']} row commitments of >= 128 rows (witness pair, derefs)")
        rk = bench.get("roofline", {})
        if "sc_cubic3_kernel<true, true>" in res and rk:
            res["sc_cubic3_kernel<true, true>"]["algorithmic_bytes_per_launch"] = rk.get("alg_bytes_per_launch")
    doc = {"_how": "rocprofv3 --pmc FETCH_SIZE and rocprofv3 --pmc WRITE_SIZE, two separate runs of the section's command, averaged per "
                   "dispatch by tools/pmc_summary.py; counters in KiB; gfx950: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
    if os.path.exists(OUT):
        doc = json.load(open(OUT))
    doc[section] = dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:24])
    json.dump(doc, open(OUT, "w"), indent=1)
    for k, v in doc[section].items():
        if k.startswith("prod_round") and "(>=" in k and bench:
            for s_ in bench.get("roofline", {}).get("secondary", []):
                if s_["kernel"].startswith("prod_round") and section != "bench_default":
                    v["algorithmic_bytes_per_launch"] = s_["alg_bytes_per_launch"]
    json.dump(doc, open(OUT, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc[section].items() if "cubic3" in k or "prod_round" in k or "msm_rows" in k}, indent=1))


if __name__ == "__main__":
    main()
