for rep in 1 2; do for r in 8 16 24 32 48 64; do
  timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --tune "{\"link_rows\": $r}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('link_rows=$r rep=$rep', d['value'], d['ms_per_step'], 'check', r['all_flooding_kernels']['per_kernel'], 'probe', r['hbm_probe_gbs'])"
done; done
