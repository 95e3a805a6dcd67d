set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND_TAG:-r01k}
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err &&
python3 bench.py --sat-only > $O/bench_sat_only.json 2> $O/bench_sat_only.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --no-verify > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err &&
python3 tools/summarize_rocprof.py $O/kt $O/${ROUND_TAG:-r01k}_rocprofv3_kernel > $O/kt_summary.txt 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 bench.py --trace L5 --only mult --sat-only --serial --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/pmc_f.json 2> $O/pmc_f.err &&
python3 tools/summarize_rocprof.py $O/pmc_f $O/${ROUND_TAG:-r01k}_rocprofv3_pmc_FETCH_SIZE > $O/pmc_f_summary.txt 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 bench.py --trace L5 --only mult --sat-only --serial --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/pmc_w.json 2> $O/pmc_w.err &&
python3 tools/summarize_rocprof.py $O/pmc_w $O/${ROUND_TAG:-r01k}_rocprofv3_pmc_WRITE_SIZE > $O/pmc_w_summary.txt 2>&1
echo rc=$?
rm -rf $O/kt $O/pmc_f $O/pmc_w
ls -la $O
