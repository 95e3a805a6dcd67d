#!/bin/bash
# same-box A/B of the default LeNet step with the row-per-lane derefs commitment off / S strips
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ab
mkdir -p $O
for S in 0 16 8 auto 0 16; do
  if [ $S = auto ]; then unset VPIN_MSM_STRIP; else export VPIN_MSM_STRIP=$S; fi
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/b_$S.json 2> $O/b_$S.err
  python3 - <<PY
import json
d=json.loads(open("$O/b_$S.json").read().strip().splitlines()[-1])
print("S=$S", round(d["ms_per_step"],1), "ms/step", d["spans_ms_last_step"]["L5-mult"]["spark_derefs_commit"], flush=True)
PY
done
