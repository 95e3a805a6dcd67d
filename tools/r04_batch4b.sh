#!/bin/bash
# round-4 final measurements: whole GPU suite, default line, per-config lines, CLI cold (process per label / one process), profiles
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b4
mkdir -p $O
cd $R
python3 bench.py --trace L5 --only mult --serial --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_L5_mult.json 2> $O/bench_L5_mult.err
python3 -c "import json; d=json.loads(open('$O/bench_L5_mult.json').read().strip().splitlines()[-1]); print('L5-mult alone:', round(d['ms_per_step'],1), 'ms', flush=True)"
python3 bench.py --sat-only --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_sat_only.json 2> $O/bench_sat_only.err
python3 -c "import json; d=json.loads(open('$O/bench_sat_only.json').read().strip().splitlines()[-1]); print('sat only:', round(d['ms_per_step'],1), 'ms', flush=True)"
python3 tools/time_cli.py --scrub --one-process L1 L2 L3 L4 L5 L6 L7 > $O/cli_one_process.txt 2>&1; grep "== one process" $O/cli_one_process.txt
python3 tools/time_cli.py --scrub L1 L2 L3 L4 L5 L6 L7 > $O/cli_process_per_label.txt 2>&1; grep "Total proof generation\|process wall" $O/cli_process_per_label.txt | tr '\n' ' '; echo
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
