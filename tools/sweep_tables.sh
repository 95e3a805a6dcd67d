#!/bin/bash
# tools/sweep_tables.sh [steps] -- the LeNet step at several window-table budgets: ms/step, J/step (median
# socket power x time), HBM in use, table bytes.  Budgets are (sat stream GB, eval stream GB); the default is 24 / 80.
# Writes one line per budget to stdout; detail files under gpurun_out/.
STEPS=${1:-20}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for B in "24 80" "16 48" "10 30" "5 15"; do
  set -- $B
  OUT=gpurun_out/tables_$1_$2.json
  VPIN_GENS_BUDGET_GB=$1 VPIN_SPARK_GENS_BUDGET_GB=$2 timeout -k 10 300 python3 bench.py --table-slot 96 --steps $STEPS --warmup 3 --no-cpu-baseline --no-live-pmc \
      --no-span --no-roofline-pass --detail-out $OUT > gpurun_out/tables_$1_$2.line 2> gpurun_out/tables_$1_$2.err || { echo "budget $1/$2 failed"; tail -3 gpurun_out/tables_$1_$2.err; continue; }
  python3 - $1 $2 gpurun_out/tables_$1_$2.line <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"budget sat {sys.argv[1]:>2} GB + eval {sys.argv[2]:>2} GB | tables {d['window_tables_gib']:6.2f} GiB | HBM in use {d['hbm_in_use_gib']:6.1f} GiB | "
      f"{d['ms_per_step']:7.2f} ms/step | {d['value'] / 1e6:6.2f} M constraints/s | {d.get('watts_median')} W | {d.get('joules_per_step')} J/step | "
      f"bytes_ok {d['bytes_ok']} verified {d['verified_ok']}")
PY
done
