#!/bin/bash
# ablation scan of one kernel family: tools/ab/dbg_scan.sh VAR family v1 v2 ...   (prints the family's ms per step for each value of VAR)
VAR=$1; FAM=$2; shift 2
for v in "$@"; do
  env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-gpu-baseline --no-small-batch --no-inference --no-c5 --no-loss-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
kb = {k['kernel']: k['ms_per_step'] for k in d['kernel_breakdown']}
print('$VAR=$v', d['ms_per_step'], '$FAM', kb.get('$FAM'))"
done
