#!/bin/bash
# BG1-profile layered configuration (configs[3]) over several builds of the library and tuning variants,
# alternating on one box: tools/ab_ldsp_libs.sh reps sigma 'lib|<tune json>' ...   (lib "" = current)
REPS=${1:-2}; SIGMA=${2:-0}; shift; shift
for rep in $(seq $REPS); do for v in "$@"; do
  lib=${v%%|*}; tune=${v#*|}
  LDPC_HIP_LIB=$lib timeout -k 10 120 python bench.py --config bg1_layered --steps 6 --warmup 2 --sigma $SIGMA --tune "$tune" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('lib=${lib:-current}', '$tune', d['value'], 'Mbit/s', d['ms_per_step'], 'ms  avg iterations', c.get('avg_iterations_per_frame'), 'converged', c.get('frames_converged'), list(d['roofline']['per_kernel']))"
done; done
