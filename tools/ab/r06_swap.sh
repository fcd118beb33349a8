#!/bin/bash
# cross-lane exchanges as VALU lane swaps (v_permlane32_swap / v_permlane16_swap) instead of ds_bpermute shuffles: the committed build (shfl) / the swaps in the
# 32x32x16 chain kernels + layer_dw (swap) / the 16x16x32 chain kernels with swaps (c16swap); same box, bench.py --steps 100
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06v
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in shfl swap c16swap; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if 'chain' in k or k=='layer_dw'})"
  done
done 2>&1 | tee gpurun_out/r06v/ab_swap.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
