#!/bin/bash
# A/B on one box with one build: alternate environment settings, 3 rounds each; prints rays/s, ms per step and the per-family kernel times.
# Usage: tools/ab/run_env2.sh "VAR=a" "VAR=b" [rays]     (use X=0 for "no switch")
R=${3:-4096}
for round in 1 2 3; do
  for v in "$1" "$2"; do
    env $v python bench.py --rays $R --steps 40 --warmup 8 --no-cpu-baseline --no-torch-gpu-baseline --no-small-batch --no-inference --no-c5 --no-loss-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
kb = {k['kernel']: k['ms_per_step'] for k in d['kernel_breakdown']}
print('$v', d['value'], d['ms_per_step'], 'launches', d.get('launches_per_step'), ' '.join('%s=%.3f' % (k, v) for k, v in sorted(kb.items(), key=lambda kv: -kv[1])[:9]))"
  done
done
