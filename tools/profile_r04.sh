#!/bin/bash
# Profiles of round 4 (run on the GPU box from the repo root): kernel trace + stats of the default bench, the SQ (VALU-issue)
# counters of the largest instance alone, and the two HBM-traffic passes (FETCH_SIZE / WRITE_SIZE do not fit one pass).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify > $O/bench_L5_mult.json 2> $O/bench_L5_mult.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-span > $O/bench_under_rocprof.json 2> $O/ktrace.log || exit 2
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_L5_valu -o p -- python3 $R/bench.py --trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/pmc_L5_valu.json 2> $O/pmc_L5_valu.log || exit 3
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_L5_$ctr -o p -- python3 $R/bench.py --trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/pmc_L5_$ctr.json 2> $O/pmc_L5_$ctr.log || exit 4
done
cd $R
VPIN_PMC_VALU_OUT=r04_pmc_valu.json python3 tools/pmc_valu.py $O/pmc_L5_valu $O/bench_L5_mult.json > $O/pmc_valu_summary.txt
VPIN_PMC_OUT=r04_pmc_traffic.json python3 tools/pmc_summary.py bench_L5_mult $O/pmc_L5_FETCH_SIZE $O/pmc_L5_WRITE_SIZE $O/bench_L5_mult.json > $O/pmc_L5_summary.txt
python3 tools/summarize_rocprof.py $O/ktrace $O/r04_rocprofv3
cp profiles/r04_pmc_valu.json profiles/r04_pmc_traffic.json $O/ 2>/dev/null
find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O
