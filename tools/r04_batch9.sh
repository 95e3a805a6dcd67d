#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b9
mkdir -p $O
cd $R
run() { name=$1; shift
  env "$@" python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass --no-live-pmc $EXTRA > $O/$name.json 2> $O/$name.err
  python3 -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name:', round(d['ms_per_step'],1), 'ms/step', flush=True)" || tail -3 $O/$name.err
}
run base X=1
run q8 GPU_MAX_HW_QUEUES=8
run q2 GPU_MAX_HW_QUEUES=2
EXTRA=--no-prof; run noprof X=1; EXTRA=
run base2 X=1
