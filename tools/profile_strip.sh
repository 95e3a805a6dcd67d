#!/bin/bash
# the 2^25 instance proven alone: row-per-lane derefs commitment with its workgroups kept in step (VPIN_MSM_STRIP_LAG generators of slack; 0 = free running)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04strip
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VPIN_MSM_STRIP_TRACE=1
for LAG in 0 1 2 4; do
  export VPIN_MSM_STRIP_LAG=$LAG
  python3 $R/bench.py --trace L5 --only mult --serial --steps 6 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/b_lag$LAG.json 2> $O/b_lag$LAG.err
  python3 -c "import json; d=json.loads(open('$O/b_lag$LAG.json').read().strip().splitlines()[-1]); print('lag=$LAG', round(d['ms_per_step'],1), 'ms/step, derefs', d['spans_ms_last_step']['L5-mult']['spark_derefs_commit'], all(d['bytes_equal_oracle_digest'].values()), flush=True)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktl$LAG -o kt -- python3 $R/bench.py --trace L5 --only mult --serial --steps 2 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/bench_l$LAG.json 2> $O/ktl$LAG.log || exit 2
  python3 - <<PY
import csv, glob
f=glob.glob("$O/ktl$LAG/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'msm_strip' in n or 'msm_rows_hot' in n:
        print('   ', n.split('(')[0][-36:], r['Calls'], 'avg ms', round(float(r['AverageNs'])/1e6,3), flush=True)
PY
  find $O/ktl$LAG -name "*kernel_trace.csv" -size +20M -delete
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_l$LAG -o p -- python3 $R/bench.py --trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/pmc_fetch_l$LAG.json 2> $O/pmc_fetch_l$LAG.log
  python3 - <<PY
import csv, glob, collections
f=glob.glob("$O/pmc_fetch_l$LAG/**/*counter_collection.csv", recursive=True)[0]
tot=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'msm_strip' in k or 'msm_rows_hot' in k:
        tot[k]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in tot: print('    FETCH_SIZE raw per dispatch', k[-30:], tot[k]/len(n[k]), flush=True)
PY
  find $O/pmc_fetch_l$LAG -name "*counter_collection.csv" -size +20M -delete
done
