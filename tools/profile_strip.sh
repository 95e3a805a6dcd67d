#!/bin/bash
# kernel trace of the 2^25 instance proven alone with the row-per-lane derefs commitment on (S = 16) and off
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04strip
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in 16 0; do
  export VPIN_MSM_STRIP=$S
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$S -o kt -- python3 $R/bench.py --trace L5 --only mult --serial --steps 2 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/bench_$S.json 2> $O/kt$S.log || exit 2
  f=$(find $O/kt$S -name "*kernel_stats.csv" | head -1)
  head -12 "$f" | cut -c1-160 > $O/stats_$S.txt
  find $O/kt$S -name "*kernel_trace.csv" -size +20M -delete
done
cat $O/stats_16.txt $O/stats_0.txt
