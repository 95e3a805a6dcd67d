#!/bin/bash
# round-4 final measurements: whole GPU suite, default line, per-config lines, CLI cold (process per label / one process), profiles
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b4
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
for T in 4 8; do
  VPIN_HOST_THREADS=$T python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/ht_$T.json 2> $O/ht_$T.err
  python3 -c "import json; d=json.loads(open('$O/ht_$T.json').read().strip().splitlines()[-1]); print('host threads $T:', round(d['ms_per_step'],1), 'ms/step', flush=True)"
done
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default:', round(d['ms_per_step'],1), 'ms/step', round(d['value']/1e6,2), 'M c/s; roofline frac', round(d['roofline']['frac'],3), '; span', d.get('reference_span',{}).get('ms_per_trace'), d.get('reference_span',{}).get('dead_work',{}).get('ms_per_trace_with'), '; cpu', round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], flush=True)"
for T in A 3_32 7_256 E; do
  python3 bench.py --trace $T --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$T.json 2> $O/bench_$T.err
  python3 -c "import json; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('$T:', round(d['ms_per_step'],2), 'ms/step', round(d['value']/1e6,2), 'M c/s', flush=True)"
done
