#!/bin/bash
# How often does a process find a fast pair of allocations?  N short bench processes, one line each:
# step time, the candidates' round times and which was kept.   tools/placement_stats.sh [N] [extra bench args]
N=${1:-12}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $N); do
  python3 $REPO/bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['roofline']['placement']
print('%2d  step %.1f ms  check %.3f ms  best %.3f of %d candidates  %s' % ($i, d['ms_per_step'], d['roofline']['avg_launch_ms'], min(p['candidates_ms']), len(p['candidates_ms']), p['candidates_ms']))
"
done
