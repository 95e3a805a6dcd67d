#!/bin/bash
# tools/check_bench.sh <file holding bench.py's stdout>: fails unless the LAST line is strict JSON of at most 4096 bytes with the
# contract's keys and the `roofline` / `cpu_baseline` objects; prints the interesting scalars.
python3 - "$1" <<'PY'
import json, sys
last = open(sys.argv[1]).read().strip().splitlines()[-1]
def bad(c):
    raise ValueError("non-finite constant " + c)
try:
    d = json.loads(last, parse_constant=bad)
except ValueError as e:
    sys.exit(f"FAIL: the last line is not strict JSON: {e}")
if len(last.encode()) > 4096:
    sys.exit(f"FAIL: the line is {len(last.encode())} bytes (> 4096)")
need = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
miss = [k for k in need if k not in d]
if miss:
    sys.exit(f"FAIL: missing {miss}")
for sub in ("roofline", "cpu_baseline"):
    if sub not in d and d["n_gpus"] == 1:
        sys.exit(f"FAIL: no {sub} object")
    for k, v in d.get(sub, {}).items():
        if isinstance(v, (dict, list)):
            sys.exit(f"FAIL: {sub}.{k} is not a scalar")
print("bytes", len(last.encode()), "| ms/step", round(d["ms_per_step"], 2), "value", round(d["value"]), "hbm", d.get("hbm_in_use_gib"), "run_s", d.get("run_s"))
print("roofline", d.get("roofline"))
print("cpu_baseline", d.get("cpu_baseline"))
print("span", d.get("reference_span_ms"), "largest", d.get("reference_span_largest_ms"), "lanes", d.get("reference_span_lanes_ms"),
      "value_reference_span", d.get("value_reference_span"), "warning", d.get("span_warning"))
print("bytes ok:", d.get("bytes_ok"), "verified:", d.get("verified_ok"), "J/step", d.get("joules_per_step"), "detail", d.get("detail"))
print({k: v for k, v in d.items() if k.startswith("strong_")})
if d.get("span_warning"):
    sys.exit("FAIL: " + d["span_warning"])
PY
