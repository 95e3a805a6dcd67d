#!/bin/bash
# default LeNet step: more lanes for the small instances now that the host threads are sized to the cores
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b6
mkdir -p $O
cd $R
run() { name=$1; shift; ht=$1; shift
  VPIN_HOST_THREADS=$ht python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass "$@" > $O/$name.json 2> $O/$name.err
  python3 -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name:', round(d['ms_per_step'],1), 'ms/step', flush=True)" || tail -5 $O/$name.err
}
run base_ht4 4
run mult3_ht3 3 --mult-lanes 3
run mult4_ht2 2 --mult-lanes 4
run six_ht2 2 --lanes-spec "L5-mult;L3-mult;L1-mult;L6-mult,L7-mult;L2-add,L5-add,L4-add;L1-add,L3-add,L6-add,L7-add"
run five_ht3 3 --lanes-spec "L5-mult;L3-mult;L1-mult,L6-mult,L7-mult;L2-add,L5-add,L4-add;L1-add,L3-add,L6-add,L7-add"
run base_ht4b 4
