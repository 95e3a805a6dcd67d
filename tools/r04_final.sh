#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -2 $O/gputest.log
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; print('default:', round(d['ms_per_step'],1), 'ms/step', round(d['value']/1e6,2), 'M c/s; roofline frac', round(r['frac'],3), r['avg_launch_us'], 'traffic', r['traffic'], r['traffic_source'][:30], '; span', d.get('reference_span',{}).get('ms_per_trace'), d.get('reference_span',{}).get('lanes',{}).get('ms_per_trace'), '; cpu', round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], '; hbm', d['hbm_in_use_gib_after_timed_region'], 'verify', d['verify_s'], all(d['bytes_equal_oracle_digest'].values()), flush=True)"
