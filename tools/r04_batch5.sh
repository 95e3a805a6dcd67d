#!/bin/bash
# default LeNet step: host threads per lane and the gate of the other lanes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b5
mkdir -p $O
cd $R
run() { name=$1; shift
  env "$@" python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass $EXTRA > $O/$name.json 2> $O/$name.err
  python3 -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name:', round(d['ms_per_step'],1), 'ms/step', d.get('bytes_equal_oracle_digest') if not isinstance(d.get('bytes_equal_oracle_digest'), dict) else all(d['bytes_equal_oracle_digest'].values()), flush=True)" || tail -5 $O/$name.err
}
EXTRA=""
run ht2 VPIN_HOST_THREADS=2
run ht3 VPIN_HOST_THREADS=3
run ht4 VPIN_HOST_THREADS=4
run ht6 VPIN_HOST_THREADS=6
run ht8 VPIN_HOST_THREADS=8
run ht4b VPIN_HOST_THREADS=4
EXTRA="--gate 2"; run ht4_gate2 VPIN_HOST_THREADS=4
EXTRA="--gate 3"; run ht4_gate3 VPIN_HOST_THREADS=4
EXTRA="--gate 2"; run ht4_gate2_strip VPIN_HOST_THREADS=4 VPIN_MSM_STRIP=16
EXTRA="--gate 1"; run ht4_gate1 VPIN_HOST_THREADS=4
