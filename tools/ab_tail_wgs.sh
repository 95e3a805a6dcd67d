#!/bin/bash
# tools/ab_tail_wgs.sh -- A/B of VPIN_SPARK_TAIL_WGS (round 6): the rounds between 1024 and 8192 pairs per circuit inside the
# resident tail kernel on 2 / 4 / 8 workgroups per circuit against a launch per round (1 = the default).  Every trace twice per
# setting, interleaved, on one box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-pmc --no-span --no-roofline-pass --no-verify"
for T in 3_32 A 7_256 E; do
  for rep in 1 2; do
    for W in 1 2 4 8; do
      VPIN_SPARK_TAIL_WGS=$W timeout -k 10 200 python3 bench.py --trace $T --steps 30 --warmup 5 $F --detail-out gpurun_out/abw.json > gpurun_out/abw.line 2> gpurun_out/abw.err || { echo "trace $T W=$W failed"; continue; }
      python3 -c "
import json;d=json.loads(open('gpurun_out/abw.line').read().strip().splitlines()[-1]);print('trace %-6s run $rep  VPIN_SPARK_TAIL_WGS=$W  %8.3f ms/step  %7.2f M constraints/s' % ('$T', d['ms_per_step'], d['value']/1e6))"
    done
  done
done
for rep in 1 2; do
  for W in 1 8; do
    VPIN_SPARK_TAIL_WGS=$W timeout -k 10 300 python3 bench.py --steps 15 --warmup 3 $F --detail-out gpurun_out/abw.json > gpurun_out/abw.line 2> gpurun_out/abw.err || { echo "lenet W=$W failed"; continue; }
    python3 -c "
import json;d=json.loads(open('gpurun_out/abw.line').read().strip().splitlines()[-1]);print('trace lenet  run $rep  VPIN_SPARK_TAIL_WGS=$W  %8.3f ms/step  %7.2f M constraints/s' % (d['ms_per_step'], d['value']/1e6))"
  done
done
