#!/bin/bash
# default LeNet step: L5-mult is the critical path (403 of 406 ms) -- how many workgroups per CU should its row commitments take on the shared device?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b8
mkdir -p $O
cd $R
for V in 1 0 1 0; do
  if [ $V = 1 ]; then export VPIN_MSM_STRIP_SHARED=1; else unset VPIN_MSM_STRIP_SHARED; fi
  VPIN_MSM_STRIP_TRACE=1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/sh_$V.json 2> $O/sh_$V.err
  python3 -c "import json; d=json.loads(open('$O/sh_$V.json').read().strip().splitlines()[-1]); s=d['spans_ms_last_step']; print('strip on the shared device $V:', round(d['ms_per_step'],1), 'ms/step; L5-mult', s['L5-mult']['spark_total'], 'derefs', s['L5-mult']['spark_derefs_commit'], flush=True)"
done
grep -h "\[strip\]" $O/sh_1.err | sort | uniq -c | head -3
