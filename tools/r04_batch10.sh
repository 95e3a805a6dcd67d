#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b10
mkdir -p $O
cd $R
for V in 0 1 0 1; do
  if [ $V = 1 ]; then export VPIN_GENS_TMP_FREE=1; else unset VPIN_GENS_TMP_FREE; fi
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass --no-live-pmc > $O/tmpfree_$V.json 2> $O/tmpfree_$V.err
  python3 -c "import json; d=json.loads(open('$O/tmpfree_$V.json').read().strip().splitlines()[-1]); print('tmp free $V:', round(d['ms_per_step'],1), 'ms/step; hbm in use', d['hbm_in_use_gib_after_timed_region'], 'GiB; setup', d['setup_s'], flush=True)" || tail -3 $O/tmpfree_$V.err
done
