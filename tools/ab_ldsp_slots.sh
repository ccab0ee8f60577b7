#!/bin/bash
# BG1-profile layered configuration (configs[3]): frames per workgroup x workgroups per CU of
# layered_ldsp_kernel, alternating on one box: tools/ab_ldsp_slots.sh reps sigma '<tune json>' ...
REPS=${1:-2}; SIGMA=${2:-0}; shift; shift
for rep in $(seq $REPS); do for v in "$@"; do
  timeout -k 10 120 python bench.py --config bg1_layered --steps 6 --warmup 2 --sigma $SIGMA --tune "$v" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$v', d['value'], 'Mbit/s', d['ms_per_step'], 'ms  avg iterations', c.get('avg_iterations_per_frame'), 'converged', c.get('frames_converged'), list(d['roofline']['per_kernel']))"
done; done
