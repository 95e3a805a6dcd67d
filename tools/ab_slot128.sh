#!/bin/bash
# tools/ab_slot128.sh -- window-table entries in 128-byte slots (one cache line per gather) against the shipped 96-byte packing, with
# round 6's schedule (spatial split, strips).  BOTH libraries built on the box.  The budgets keep 12-bit windows for the 16386
# generators the derefs commitments walk in both layouts (128 B: 94.5 GB; 96 B: 70.9 GB).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export VPIN_SPARK_GENS_BUDGET_GB=100 VPIN_GENS_BUDGET_GB=24 VPIN_GENS_FREE_FRACTION=0.45
F="--no-cpu-baseline --no-live-pmc --no-span --no-verify --no-roofline-pass"
run() {
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --steps 15 --warmup 3 $F --detail-out gpurun_out/sl.json > gpurun_out/sl.line 2> gpurun_out/sl.err || { echo "$1 failed"; tail -3 gpurun_out/sl.err; continue; }
    python3 -c "
import json;d=json.load(open('gpurun_out/sl.json'));s=d['spans_ms_last_step']['L5-mult']
print('$1 run $rep  LeNet step %.2f ms | L5-mult derefs commit %.1f ms, polycommit %.1f | tables %.1f GiB, HBM in use %.1f GiB | %s W' % (d['ms_per_step'], s['spark_derefs_commit'], s['polycommit'], d['hbm_breakdown']['window_tables_gib'], d['hbm_in_use_gib_after_timed_region'], d['power_during_timed_region']['watts_median']))"
  done
}
run "slot  96 (shipped)"
touch vpin_amd/csrc/msm.hip
VPIN_HIPCC_FLAGS="-DVPIN_NIELS_SLOT=128" python3 -c "from vpin_amd import build; build.build()" || exit 1
run "slot 128          "
touch vpin_amd/csrc/msm.hip
python3 -c "from vpin_amd import build; build.build()" || exit 1
run "slot  96 again    "
