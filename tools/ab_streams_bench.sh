#!/bin/bash
# headline bench with one and with two streams, alternating on one box: tools/ab_streams_bench.sh reps
REPS=${1:-3}
for rep in $(seq $REPS); do for st in 0 2; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('streams $st', d['value'], d['ms_per_step'], 'dominant', r['kernel'], r['avg_launch_ms'], 'frac', r['frac'], 'of probe', r['frac_of_probe'], {k:v['avg_ms'] for k,v in r['all_flooding_kernels']['per_kernel'].items()}, 'whole', r['whole_step_frac'])"
done; done
