#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04reh
mkdir -p $O
cd $R
for W in 8 4 2; do
  timeout -k 10 500 python3 bench.py --scaling strong --rehearse $W --steps 2 --n1-step-ms 400 > $O/rehearse$W.json 2> $O/rehearse$W.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/rehearse$W.json").read().strip().splitlines()[-1])
    i=d["instances"]["L5-mult"]
    print("W=$W: trace model", d["model_ms"], "ms (lower bound", d["model_ms_lower_bound_quietest_pass"], "), vs serial", d["model_speedup_vs_single_gpu_serial"], "vs N=1 step", d["model_speedup_vs_n1_four_lane_step"], "; L5-mult", i.get("single_gpu_ms"), "->", i.get("model_ms"), i.get("fraction_of_single_gpu"), i.get("fraction_of_single_gpu_quietest"), flush=True)
except Exception as e:
    print("W=$W ERR", e, open("$O/rehearse$W.err").read()[-600:], flush=True)
PY
done
