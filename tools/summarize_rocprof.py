"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.
usage: summarize_rocprof.py <rocprof_out_dir> <out_prefix>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    name = name.replace("vpin::", "")
    return name.split("(")[0][:70]


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    ks = find(d, "*kernel_stats.csv")
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        res["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]),
                                "total_ns": int(float(r["TotalDurationNs"])), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in rows[:25]]
        with open(out + "_kernel_stats.csv", "w") as f:
            f.write(open(ks[0]).read())
    kt = find(d, "*kernel_trace.csv")
    if kt and "kernel_stats" not in res:
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(kt[0])):
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += dur
        tot = sum(v[1] for v in agg.values())
        res["kernel_stats"] = sorted(({"name": k, "calls": v[0], "total_ns": v[1], "avg_ns": v[1] / v[0],
                                       "pct": 100 * v[1] / tot} for k, v in agg.items()), key=lambda x: -x["total_ns"])[:25]
    if kt:
        # several HIP streams run concurrently in bench.py's default schedule: a kernel's duration on one
        # stream then includes the time it shares the CUs with the other streams' kernels, so the per-stream
        # breakdown is what compares with bench.py's per-lane HIP-event numbers
        by = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(kt[0])):
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a = by[(short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?")))]
            a[0] += 1
            a[1] += dur
        top = {x["name"] for x in res.get("kernel_stats", [])[:12]} | {k[0] for k in by if "sc_cubic3" in k[0]}  # + the roofline kernel
        res["kernel_stats_by_stream"] = sorted(({"name": k[0], "stream": k[1], "calls": v[0], "total_ns": v[1], "avg_ns": v[1] / v[0]}
                                                for k, v in by.items() if k[0] in top), key=lambda x: (x["name"], x["stream"]))
    cc = find(d, "*counter_collection.csv")
    if cc:
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(cc[0])):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        res["counters"] = {k: {cn: {"dispatches": v[0], "sum": v[1], "avg": v[1] / v[0]} for cn, v in cs.items()}
                           for k, cs in agg.items()}
    with open(out + "_summary.json", "w") as f:
        json.dump(res, f, indent=1)
    for r in res.get("kernel_stats", [])[:12]:
        print(f'{r["pct"]:6.2f}%  calls={r["calls"]:5d}  avg={r["avg_ns"] / 1e3:9.2f} us  {r["name"]}')
    for r in res.get("kernel_stats_by_stream", []):
        if "sc_cubic3" in r["name"] or "prod_round" in r["name"]:
            print(f'   stream {r["stream"]}: calls={r["calls"]:5d} avg={r["avg_ns"] / 1e3:9.2f} us  {r["name"]}')
    for k, cs in res.get("counters", {}).items():
        print(k, {cn: round(v["avg"], 1) for cn, v in cs.items()})


if __name__ == "__main__":
    main()
