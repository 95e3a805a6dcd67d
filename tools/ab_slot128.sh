#!/bin/bash
# A/B on the box: window-table entries in 128-byte slots (one cache line per gather) against the shipped 96-byte packing.
# Builds the library twice ON THE BOX (the tree's lib is the 96-byte one); budgets raised so that both keep 12-bit windows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04slot
mkdir -p $O
cd $R
export VPIN_SPARK_GENS_BUDGET_GB=100 VPIN_GENS_BUDGET_GB=24 VPIN_GENS_FREE_FRACTION=0.45
for SLOT in 128 96; do
  VPIN_HIPCC_FLAGS="-DVPIN_NIELS_SLOT=$SLOT" python3 -c "from vpin_amd import build; build.build(force=True)" > $O/build_$SLOT.log 2>&1 || { tail -5 $O/build_$SLOT.log; exit 1; }
  python3 bench.py --trace L5 --only mult --serial --steps 6 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/l5_$SLOT.json 2> $O/l5_$SLOT.err
  python3 -c "import json; d=json.loads(open('$O/l5_$SLOT.json').read().strip().splitlines()[-1]); s=d['spans_ms_last_step']['L5-mult']; print('slot $SLOT: L5-mult alone', round(d['ms_per_step'],1), 'ms; derefs', s['spark_derefs_commit'], 'polycommit', s['polycommit'], 'hbm GiB', d['hbm_in_use_gib_after_timed_region'], all(d['bytes_equal_oracle_digest'].values()), flush=True)" || tail -3 $O/l5_$SLOT.err
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/def_$SLOT.json 2> $O/def_$SLOT.err
  python3 -c "import json; d=json.loads(open('$O/def_$SLOT.json').read().strip().splitlines()[-1]); print('slot $SLOT: default step', round(d['ms_per_step'],1), 'ms; hbm GiB', d['hbm_in_use_gib_after_timed_region'], all(d['bytes_equal_oracle_digest'].values()), flush=True)" || tail -3 $O/def_$SLOT.err
done
