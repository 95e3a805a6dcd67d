#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b12
mkdir -p $O
cd $R
for T in 8 16 4 8; do
  VPIN_HOST_THREADS=$T python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/l5_ht$T.json 2> $O/l5_ht$T.err
  python3 -c "import json; d=json.loads(open('$O/l5_ht$T.json').read().strip().splitlines()[-1]); print('L5 alone, host threads $T:', round(d['ms_per_step'],1), flush=True)"
done
VPIN_HOST_THREADS=8 python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass --no-prof > $O/l5_noprof.json 2>/dev/null
python3 -c "import json; d=json.loads(open('$O/l5_noprof.json').read().strip().splitlines()[-1]); print('L5 alone, no prof:', round(d['ms_per_step'],1), flush=True)"
