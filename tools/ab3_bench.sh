#!/bin/bash
# alternate N tuning variants of the headline benchmark on one box: tools/ab3_bench.sh reps '<json>' '<json>' ...
REPS=$1; shift
for rep in $(seq $REPS); do for v in "$@"; do
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --tune "$v" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in r['all_flooding_kernels']['per_kernel'].items()}, 'probe', r['hbm_probe_gbs'])"
done; done
