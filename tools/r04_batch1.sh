#!/bin/bash
# round-4 batch: new parity tests, concurrent small traces, tail-poll s_sleep A/B, two-rank rehearsal of the default N>1 line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b1
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_config_variants.py tests/test_cli.py tests/test_gpu_gadget_dev.py tests/test_gpu_gadget_pins.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for T in 3_32 A 7_256; do
  python3 bench.py --trace $T --concurrent 1,2,4,8,12,16 --steps 6 --warmup 2 > $O/concurrent_$T.json 2> $O/concurrent_$T.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/concurrent_$T.json").read().strip().splitlines()[-1])
    print("$T single", round(d["single_trace"]["ms"],2), "ms;", [(r["K"], round(r["constraints_per_s"]/1e6,2), round(r["x_single_trace_rate"],2)) for r in d["concurrent"]], d["bytes_equal_oracle_digest"], flush=True)
except Exception as e:
    print("$T ERR", e, open("$O/concurrent_$T.err").read()[-800:], flush=True)
PY
done
for V in 0 1 0 1; do
  if [ $V = 1 ]; then export VPIN_TAIL_SLEEP=1; else unset VPIN_TAIL_SLEEP; fi
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-span --no-verify --no-roofline-pass > $O/sleep_$V.json 2> $O/sleep_$V.err
  python3 -c "import json; d=json.loads(open('$O/sleep_$V.json').read().strip().splitlines()[-1]); print('tail s_sleep=$V', round(d['ms_per_step'],1), 'ms/step', flush=True)"
done
unset VPIN_TAIL_SLEEP
# the default command at N = 2 with both ranks on this one GPU (gloo: RCCL refuses two ranks per GPU): the strong sub-record
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --trace A --coop-log2 20 --steps 3 --warmup 1 --no-cpu-baseline --no-span --no-verify > $O/n2_gloo_A.json 2> $O/n2_gloo_A.err
tail -c 1500 $O/n2_gloo_A.json
