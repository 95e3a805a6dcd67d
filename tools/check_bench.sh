#!/bin/bash
# print the interesting scalars of a bench line: tools/check_bench.sh <file>
python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 2), "value", round(d["value"]), "hbm", d.get("hbm_in_use_gib_after_timed_region"))
rf = d.get("roofline", {})
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in rf.items() if not isinstance(v, (dict, list, str))})
print("limiter:", rf.get("limiter_frac_source"), "| traffic:", (rf.get("traffic_source") or "")[:60])
rs = d.get("reference_span", {})
dw = rs.get("dead_work", {})
print("span", rs.get("ms_per_trace"), "lanes", rs.get("lanes", {}).get("ms_per_trace"), "digest_ms", dw.get("digest_ms"), "with", dw.get("ms_per_trace_with"))
print("affinity", d.get("host_affinity"), "cpu", d.get("cpu_baseline", {}).get("value"))
ok = d.get("bytes_equal_oracle_digest")
print("bytes ok:", all(ok.values()) if isinstance(ok, dict) else ok, "verified:", all(d.get("verified", {}).values()) if d.get("verified") else None)
print({k: v for k, v in d.items() if k.startswith("strong_")})
PY
