i=0
for spec in "L5-mult;L3-mult,L1-mult,L6-mult,L7-mult;adds" "L5-mult;L7-mult,L6-mult,L1-mult,L3-mult;adds" "L5-mult;L1-mult,L6-mult,L7-mult,L3-mult;adds" "L5-mult;L1-mult,L3-mult,L6-mult,L7-mult;adds" "L5-mult;L7-mult,L6-mult,L1-mult,L3-mult;L7-add,L6-add,L3-add,L4-add,L5-add,L2-add,L1-add"; do
  i=$((i+1))
  python bench.py --lanes-spec "$spec" --no-cpu-baseline --no-verify --no-span > gpurun_out/lanes_$i.json 2> gpurun_out/lanes_$i.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/lanes_$i.json'))
print('$spec', round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {n:(round(v['total']),round(v['spark_total'])) for n,v in d['spans_ms_last_step'].items() if 'mult' in n})"
done
