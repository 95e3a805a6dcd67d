#!/bin/bash
# same-box A/B of the lazily reduced round sums (LeadAcc): both libraries built on the box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04lazy
mkdir -p $O
cd $R
for V in lazy eager lazy eager; do
  if [ $V = eager ]; then F="-DVPIN_NO_LAZY_ACC"; else F=""; fi
  VPIN_HIPCC_FLAGS="$F" python3 -c "from vpin_amd import build; build.build(force=True)" > $O/build_$V.log 2>&1 || { tail -5 $O/build_$V.log; exit 1; }
  python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline --no-span --no-verify > $O/l5_$V.json 2> $O/l5_$V.err
  python3 -c "import json; d=json.loads(open('$O/l5_$V.json').read().strip().splitlines()[-1]); s=d['spans_ms_last_step']['L5-mult']; r=d['roofline']; k=d['kernels']; print('$V: L5 alone', round(d['ms_per_step'],1), 'product layer', s['spark_product_layer'], 'roofline frac', round(r['frac'],3), round(r['avg_launch_us'],1), 'us; big rounds ms', round(k['spark_round_big']['ms']/10,2), 'all rounds', round(k['spark_round']['ms']/10,2), flush=True)"
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-live-pmc --no-roofline-pass > $O/def_$V.json 2> $O/def_$V.err
  python3 -c "import json; d=json.loads(open('$O/def_$V.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$V: default', round(d['ms_per_step'],1), 'roofline frac', round(r['frac'],3), round(r['avg_launch_us'],1), flush=True)"
done
