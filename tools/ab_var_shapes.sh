for t in '{}' '{"tiles_first":true}' '{"cols_per_wave":2}' '{"cols_per_wave":4}' '{"tiles_first":true,"cols_per_wave":2}' '{}'; do
  python3 bench.py --no-extras --no-cpu-baseline --steps 4 --tune "$t" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$t'.ljust(44), d['ms_per_step'], {k.split('<')[0]:v['avg_ms'] for k,v in d['roofline']['all_flooding_kernels']['per_kernel'].items()}, min(d['roofline']['placement']['candidates_ms']))"
done
