#!/bin/bash
# A/B of two builds of the library on one box (headline benchmark): tools/ab_lib.sh <other.so> [reps] [extra bench args]
OTHER=$1; REPS=${2:-3}; shift; shift
for rep in $(seq $REPS); do for lib in "" "$OTHER"; do
  LDPC_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
pk = r.get('all_flooding_kernels',{}).get('per_kernel') or r.get('per_kernel')
print('lib=${lib:-current}', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in pk.items()}, 'probe', r.get('hbm_probe_gbs'))"
done; done
