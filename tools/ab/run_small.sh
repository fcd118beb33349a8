#!/bin/bash
# A/B on one box: alternate the two builds, 3 rounds each; prints the step rate at 4096 rays and at 512 / 1024 rays per step
for round in 1 2 3; do
  for v in base new; do
    cp tools/ab/libcolorneus_hip_$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v', d['value'], d['small_batch']['rays_per_step_per_gpu'])"
  done
done
