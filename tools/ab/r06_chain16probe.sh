#!/bin/bash
# timing probe: chain kernels' MFMAs issued as v_mfma_f32_16x16x32_f16 (wrong results, same FLOP / registers) against the product build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
cp /tmp/lib_default.so tools/ab/libs/default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in default chain16probe; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], b.get('loss'), {k:fam[k] for k in fam if 'chain' in k})"
  done
done 2>&1 | tee gpurun_out/r06p/ab_chain16probe.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
