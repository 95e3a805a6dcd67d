#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04cfg
mkdir -p $O
cd $R
for T in A 3_32 7_256 E; do
  python3 bench.py --trace $T --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$T.json 2> $O/bench_$T.err
  python3 -c "import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$T:', round(d['ms_per_step'],2), 'ms/step', round(d['value']/1e6,2), 'M c/s', d.get('power_during_timed_region',{}).get('sclk_mhz_median'), d.get('power_during_timed_region',{}).get('watts_median'), flush=True)"
done
python3 bench.py --sat-only --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_sat_only.json 2> $O/bench_sat_only.err
python3 -c "import json; d=json.loads(open('$O/bench_sat_only.json').read().strip().splitlines()[-1]); print('sat only:', round(d['ms_per_step'],1), 'ms', flush=True)"
