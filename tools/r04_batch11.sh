#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b11
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_sumcheck.py tests/test_gpu_spark.py tests/test_gpu_sat.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config_snark_bytes" >> $O/tests.log 2>&1; tail -2 $O/tests.log
python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline --no-span --no-verify > $O/l5.json 2> $O/l5.err
python3 -c "import json; d=json.loads(open('$O/l5.json').read().strip().splitlines()[-1]); s=d['spans_ms_last_step']['L5-mult']; r=d['roofline']; print('L5 alone', round(d['ms_per_step'],1), 'product layer', s['spark_product_layer'], 'sat', s['spark_sat'], 'roofline frac', round(r['frac'],3), r['avg_launch_us'], [ (x['kernel'][:12], round(x['achieved'],1)) for x in r['secondary']], flush=True)"
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-live-pmc > $O/def.json 2> $O/def.err
python3 -c "import json; d=json.loads(open('$O/def.json').read().strip().splitlines()[-1]); r=d['roofline']; print('default', round(d['ms_per_step'],1), 'roofline frac', round(r['frac'],3), r['avg_launch_us'], all(d['bytes_equal_oracle_digest'].values()), flush=True)"
