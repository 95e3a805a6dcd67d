#!/bin/bash
# Sweep of the default bench's scheduling knobs on one box (development aid): lanes for the smaller point-mult instances,
# LDS padding of the row-commitment kernels under a shared device, stream priorities.  Usage: knob_sweep.sh "TAG ENV=VAL" ...
out=gpurun_out/knobs.txt
: > $out
run() {
  local tag="$1"; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-verify --no-span --steps 12 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],1))")
  echo "$tag $ms" | tee -a $out
}
for spec in "$@"; do
  set -- $spec
  tag=$1; shift
  run "$tag" "${@:-A=1}"
done
