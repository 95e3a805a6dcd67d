#!/bin/bash
# tools/ab_minwaves.sh -- A/B of VPIN_SC_MIN_WAVES (waves per SIMD asked of the register allocator for the sum-check / product-round
# kernels: 3 = 131-140 VGPRs without scratch, 4 = 128 VGPRs and 20-48 bytes of scratch per lane), BOTH libraries built on the box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F="--no-cpu-baseline --no-live-pmc --no-span --no-verify"
run() {  # $1 = tag
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 $F --detail-out gpurun_out/mw.json > gpurun_out/mw.line 2> gpurun_out/mw.err || { echo "$1 L5 failed"; tail -3 gpurun_out/mw.err; continue; }
    python3 -c "
import json;d=json.load(open('gpurun_out/mw.json'));r=d['roofline'];k=d['kernels']
print('$1 run $rep  L5-mult alone %.2f ms | sc_cubic3 frac %.4f (%.1f us) | prod rounds >= 2^20 pairs frac %s | spark_round %.2f ms, spark_round_big %.2f ms per proof' % (d['ms_per_step'], r['frac'], r['avg_launch_us'], r.get('prod_round_frac'), k['spark_round']['ms']/d['steps'], k.get('spark_round_big',{'ms':0})['ms']/d['steps']))"
  done
  timeout -k 10 300 python3 bench.py --steps 15 --warmup 3 $F --no-roofline-pass --detail-out gpurun_out/mw.json > gpurun_out/mw.line 2> gpurun_out/mw.err && python3 -c "
import json;d=json.load(open('gpurun_out/mw.json'));print('$1        LeNet step %.2f ms, bytes ok %s' % (d['ms_per_step'], all(d['bytes_equal_oracle_digest'].values()) if d.get('bytes_equal_oracle_digest') else None))"
}
run "MIN_WAVES=3 (shipped)"
touch vpin_amd/csrc/sc_dev.h
VPIN_HIPCC_FLAGS="-DVPIN_SC_MIN_WAVES=4" python3 -c "from vpin_amd import build; build.build()" || exit 1
run "MIN_WAVES=4          "
touch vpin_amd/csrc/sc_dev.h
python3 -c "from vpin_amd import build; build.build()" || exit 1
run "MIN_WAVES=3 again    "
