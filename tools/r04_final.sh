#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default:', round(d['ms_per_step'],1), 'ms/step', round(d['value']/1e6,2), 'M c/s; roofline frac', round(d['roofline']['frac'],3), d['roofline']['avg_launch_us'], '; span', d.get('reference_span',{}).get('ms_per_trace'), d.get('value_reference_span'), d.get('reference_span',{}).get('lanes',{}).get('ms_per_trace'), '; cpu', round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], '; hbm', d['hbm_in_use_gib_after_timed_region'], 'verify', d['verify_s'], all(d['bytes_equal_oracle_digest'].values()), flush=True)"
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_L5_mult.json 2> $O/bench_L5_mult.err
python3 -c "import json; d=json.loads(open('$O/bench_L5_mult.json').read().strip().splitlines()[-1]); print('L5-mult alone:', round(d['ms_per_step'],1), 'ms', d['roofline']['secondary'][0]['achieved'], flush=True)"
