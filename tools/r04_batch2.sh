#!/bin/bash
# concurrency of small traces: what limits K?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b2
mkdir -p $O
cd $R
run() { # name, env..., K
  name=$1; shift; K=$1; shift
  env "$@" python3 bench.py --trace A --concurrent $K --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", "single", round(d["single_trace"]["ms"],1), [(r["K"], round(r["constraints_per_s"]/1e6,2), round(r["x_single_trace_rate"],2)) for r in d["concurrent"]], flush=True)
except Exception as e:
    print("$name ERR", e, open("$O/$name.err").read()[-600:], flush=True)
PY
}
run k4_t4 4 VPIN_HOST_THREADS=4
run k4_t2 4 VPIN_HOST_THREADS=2
run k4_q8 4 VPIN_HOST_THREADS=4 GPU_MAX_HW_QUEUES=8
run k8_q16 8 VPIN_HOST_THREADS=2 GPU_MAX_HW_QUEUES=16
run k8_q16_notail 8 VPIN_HOST_THREADS=2 GPU_MAX_HW_QUEUES=16 VPIN_SPARK_TAIL_PAIRS=0
run k3_t5 3 VPIN_HOST_THREADS=5
# K processes instead of K threads
for K in 2 4; do
  t0=$(date +%s.%N)
  for i in $(seq 1 $K); do
    VPIN_HOST_THREADS=$((16 / K)) python3 bench.py --trace A --serial --steps 20 --warmup 3 --no-cpu-baseline --no-span --no-verify --no-roofline-pass --no-prof > $O/proc_${K}_$i.json 2> $O/proc_${K}_$i.err &
  done
  wait
  python3 - <<PY
import json
tot=0
for i in range(1,$K+1):
    d=json.loads(open("$O/proc_${K}_%d.json"%i).read().strip().splitlines()[-1]); tot+=d["value"]; ms=d["ms_per_step"]
print("processes K=$K: sum of values", round(tot/1e6,2), "M c/s, last ms/step", round(ms,1), flush=True)
PY
done
