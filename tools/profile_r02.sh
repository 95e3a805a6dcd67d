#!/bin/bash
# Profiles of round 2 (run on the GPU box from the repo root): kernel trace + stats of the default bench, and two
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE: they do not fit one pass) of the default bench and of L5-mult alone.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify > $O/bench_L5_mult.json 2> $O/bench_L5_mult.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-span > $O/bench_under_rocprof.json 2> $O/ktrace.log || exit 2
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_default_$ctr -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/pmc_default_$ctr.json 2> $O/pmc_default_$ctr.log || exit 3
  timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_L5_$ctr -o p -- python3 $R/bench.py --trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/pmc_L5_$ctr.json 2> $O/pmc_L5_$ctr.log || exit 4
done
cd $R
python3 tools/pmc_summary.py bench_default $O/pmc_default_FETCH_SIZE $O/pmc_default_WRITE_SIZE $O/bench_under_rocprof.json > $O/pmc_default_summary.txt
python3 tools/pmc_summary.py bench_L5_mult $O/pmc_L5_FETCH_SIZE $O/pmc_L5_WRITE_SIZE $O/bench_L5_mult.json > $O/pmc_L5_summary.txt
python3 tools/summarize_rocprof.py $O/ktrace $O/r02_rocprofv3
cp profiles/r02_pmc_traffic.json $O/
# the raw per-dispatch tables are large: keep the stats and the summaries
find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O
