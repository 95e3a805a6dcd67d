#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b3
mkdir -p $O
cd $R
for T in A 3_32 7_256; do
  python3 bench.py --trace $T --concurrent 2,4,6,8,12,16 --steps 8 --warmup 2 > $O/concurrent_${T}.json 2> $O/concurrent_${T}.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/concurrent_${T}.json").read().strip().splitlines()[-1])
    print("$T single", round(d["single_trace"]["ms"],2), "ms", round(d["single_trace"]["constraints_per_s"]/1e6,2), "M c/s;", [(r["K"], round(r["constraints_per_s"]/1e6,2), round(r["x_single_trace_rate"],2), round(r["ms_per_trace_under_load"],1)) for r in d["concurrent"]], d["bytes_equal_oracle_digest"], flush=True)
except Exception as e:
    print("$T ERR", e, open("$O/concurrent_${T}.err").read()[-800:], flush=True)
PY
done
