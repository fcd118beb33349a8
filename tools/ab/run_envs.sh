#!/bin/bash
# one build, several settings of an environment switch, 2 rounds.  Usage: tools/ab/run_envs.sh VAR=a VAR=b VAR=c ...
for round in 1 2; do
  for v in "$@"; do
    env $v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-torch-gpu-baseline --no-small-batch 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
kb = {k['kernel']: k['ms_per_step'] for k in d['kernel_breakdown']}
print('$v', d['value'], d['ms_per_step'], 'ws', kb.get('layer_gemm_ws'), 'dw', kb.get('dw_gemm_hx'))"
  done
done
