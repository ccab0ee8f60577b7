#!/bin/bash
# alternate several builds of the library on one box (headline benchmark): tools/ab_libs.sh reps lib1.so lib2.so ... ("" = current)
REPS=$1; shift
for rep in $(seq $REPS); do for lib in "$@"; do
  LDPC_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lib=${lib:-current}', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in r['all_flooding_kernels']['per_kernel'].items()}, 'probe', r['hbm_probe_gbs'])"
done; done
