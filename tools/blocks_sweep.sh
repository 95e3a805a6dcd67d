# workgroup-count sweep of the round kernels under the default 3-lane schedule and for a lone L5-mult proof
for cfg in "2 512 2048" "8 512 2048" "8 128 2048" "8 64 1024" "8 32 1024" "8 64 512"; do set -- $cfg
  export VPIN_SPARK_PAIRS_PER_THREAD=$1 VPIN_ROUND_BLOCKS=$2 VPIN_SC_BLOCKS=$3
  python bench.py --no-cpu-baseline --no-verify --no-span > gpurun_out/blk.json 2>/dev/null
  python bench.py --trace L5 --only mult --serial --no-cpu-baseline --no-verify --no-span > gpurun_out/blk_b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/blk.json')); b=json.load(open('gpurun_out/blk_b.json'))
print('pairs_per_thread=$1 round_blocks=$2 sc_blocks=$3 lenet', round(d['ms_per_step'],1), round(d['roofline']['frac'],3), 'L5-mult alone', round(b['ms_per_step'],1), round(b['roofline']['frac'],3))"
done
