#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b7
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "row_per_lane or L5" > $O/tests.log 2>&1; tail -3 $O/tests.log
export VPIN_MSM_STRIP_TRACE=1
for V in on off on off; do
  if [ $V = off ]; then export VPIN_MSM_STRIP=0; else unset VPIN_MSM_STRIP; fi
  python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/l5_$V.json 2> $O/l5_$V.err
  python3 -c "import json; d=json.loads(open('$O/l5_$V.json').read().strip().splitlines()[-1]); print('L5 alone, strip $V:', round(d['ms_per_step'],1), 'ms/step, derefs', d['spans_ms_last_step']['L5-mult']['spark_derefs_commit'], flush=True)"
done
grep -h "\[strip\]" $O/l5_on.err | head -2
unset VPIN_MSM_STRIP
for G in 1 2 1 2; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass --gate $G > $O/gate_$G.json 2> $O/gate_$G.err
  python3 -c "import json; d=json.loads(open('$O/gate_$G.json').read().strip().splitlines()[-1]); print('default step, gate $G:', round(d['ms_per_step'],1), 'ms/step', all(d['bytes_equal_oracle_digest'].values()) if isinstance(d.get('bytes_equal_oracle_digest'),dict) else d.get('bytes_equal_oracle_digest'), flush=True)"
done
