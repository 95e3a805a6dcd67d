for rb in 512 2048 4096; do for sb in 2048 16384; do
  VPIN_ROUND_BLOCKS=$rb VPIN_SC_BLOCKS=$sb python bench.py --no-cpu-baseline --no-verify --no-span > gpurun_out/blk_${rb}_${sb}.json 2> gpurun_out/blk.err
  python -c "
import json
d=json.load(open('gpurun_out/blk_${rb}_${sb}.json'))
print('round_blocks=$rb sc_blocks=$sb', round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {n:(round(v['total']),round(v['spark_total'])) for n,v in d['spans_ms_last_step'].items() if 'mult' in n})"
done; done
