#!/bin/bash
# The ONE recipe behind profiles/rNN_*: run on the GPU box from the repo root (gpurun), ROUND_TAG=r06 tools/profile_round.sh [part ...]
# parts: bench configs hostbuf ktrace pmc (default: all).  Round 6: bench.py prints ONE compact line (-> *.line); the full record goes
# to --detail-out (-> *.json, what the summarising tools read).  Raw profiler output stays under gpurun_out/$ROUND_TAG/ (scratch);
# the summaries are copied by hand into profiles/ (see profiles/README.md).  Same-box A/B runs: tools/ab.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${ROUND_TAG:-r06}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
PARTS=${@:-bench configs hostbuf ktrace pmc}
Q="--no-cpu-baseline --no-span --no-live-pmc"
for part in $PARTS; do
case $part in
bench)
  # the driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5): the ONE line -> .line, the full record -> .json
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/${T}_bench_default.json > $O/${T}_bench_default.line 2> $O/bench_default.err || exit 1
  tools/check_bench.sh $O/${T}_bench_default.line || exit 1 ;;
configs)
  for L in 3_32 A 7_256 E; do
    python3 bench.py --trace $L --steps 20 --warmup 3 $Q --no-roofline-pass --detail-out $O/${T}_bench_$L.json > $O/${T}_bench_$L.line 2> $O/bench_$L.err || exit 2
  done
  python3 bench.py --trace L5 --only mult --serial --steps 10 --warmup 2 $Q --detail-out $O/${T}_bench_L5_mult.json > $O/${T}_bench_L5_mult.line 2> $O/bench_L5.err || exit 2
  python3 bench.py --sat-only --steps 10 --warmup 2 $Q --no-roofline-pass --detail-out $O/${T}_bench_sat_only.json > $O/${T}_bench_sat_only.line 2> $O/bench_sat.err || exit 2
  python3 bench.py --cu-split none --steps 10 --warmup 2 $Q --no-roofline-pass --detail-out $O/${T}_bench_default_no_cu_split.json > $O/${T}_bench_default_no_cu_split.line 2> $O/bench_nosplit.err || exit 2
  for f in 3_32 A 7_256 E L5_mult sat_only default_no_cu_split; do python3 -c "import json;d=json.load(open('$O/${T}_bench_$f.json'));print('$f', round(d['ms_per_step'],2), 'ms/step', round(d['value']/1e6,2), 'M constraints/s')"; done ;;
hostbuf)
  # the host-triplet seam, PCIe-inclusive: triplets + assignments cross the bus, CSR/CSC (+ SNARK::encode) per proof
  python3 bench.py --trace A --sat-only --host-buffers --steps 10 --warmup 3 $Q --no-roofline-pass --detail-out $O/${T}_bench_A_hostbuf_sat.json > $O/${T}_bench_A_hostbuf_sat.line 2> $O/hb1.err || exit 3
  python3 bench.py --trace A --sat-only --host-buffers --pin-host-buffers --steps 10 --warmup 3 $Q --no-roofline-pass --detail-out $O/${T}_bench_A_hostbuf_sat_pinned.json > $O/${T}_bench_A_hostbuf_sat_pinned.line 2> $O/hb2.err || exit 3
  for L in A E L5; do
    python3 bench.py --trace $L --host-buffers --pin-host-buffers --steps 5 --warmup 2 $Q --no-roofline-pass --detail-out $O/${T}_bench_${L}_hostbuf_snark.json > $O/${T}_bench_${L}_hostbuf_snark.line 2> $O/hb_$L.err || exit 3
  done
  for f in A_hostbuf_sat A_hostbuf_sat_pinned A_hostbuf_snark E_hostbuf_snark L5_hostbuf_snark; do python3 -c "import json;d=json.load(open('$O/${T}_bench_$f.json'));print('$f', round(d['ms_per_step'],2), 'ms/step')"; done ;;
ktrace)
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-span --detail-out $O/${T}_bench_default_under_rocprof.json > $O/${T}_bench_default_under_rocprof.line 2> $O/ktrace.log || exit 4
  cd $R
  python3 tools/summarize_rocprof.py $O/ktrace $O/${T}_rocprofv3 > $O/ktrace_summary.txt 2>&1
  find $O -name "*kernel_trace.csv" -size +20M -delete ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  P="--trace L5 --only mult --serial --steps 1 --warmup 1 --no-cpu-baseline --no-span --no-verify --no-live-pmc"
  python3 $R/bench.py $P --detail-out $O/bench_L5_mult_for_pmc.json > $O/bench_L5_mult_for_pmc.line 2> $O/pmc_bench.err || exit 5
  timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_L5_valu -o p -- python3 $R/bench.py $P --no-roofline-pass --detail-out $O/pmc_L5_valu.json > $O/pmc_L5_valu.line 2> $O/pmc_L5_valu.log || exit 5
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_L5_$ctr -o p -- python3 $R/bench.py $P --no-roofline-pass --detail-out $O/pmc_L5_$ctr.json > $O/pmc_L5_$ctr.line 2> $O/pmc_L5_$ctr.log || exit 5
  done
  cd $R
  VPIN_PMC_VALU_OUT=${T}_pmc_valu.json python3 tools/pmc_valu.py $O/pmc_L5_valu $O/bench_L5_mult_for_pmc.json > $O/pmc_valu_summary.txt
  VPIN_PMC_OUT=${T}_pmc_traffic.json python3 tools/pmc_summary.py bench_L5_mult $O/pmc_L5_FETCH_SIZE $O/pmc_L5_WRITE_SIZE $O/bench_L5_mult_for_pmc.json > $O/pmc_L5_summary.txt
  cp profiles/${T}_pmc_valu.json profiles/${T}_pmc_traffic.json $O/ 2>/dev/null
  find $O -name "*counter_collection.csv" -size +20M -delete ;;
esac
done
ls $O | head -60
