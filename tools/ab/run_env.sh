#!/bin/bash
# A/B on one box with one build: alternate an environment switch, 3 rounds each.  Usage: tools/ab/run_env.sh VAR=off_value VAR=on_value
for round in 1 2 3; do
  for v in "$1" "$2"; do
    env $v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-torch-gpu-baseline --no-small-batch 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
kb = {k['kernel']: k['ms_per_step'] for k in d['kernel_breakdown']}
print('$v', d['value'], d['ms_per_step'], 'ws', kb.get('layer_gemm_ws'), 'dw', kb.get('dw_gemm_hx'))"
  done
done
