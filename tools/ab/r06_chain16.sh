#!/bin/bash
# chain-fused kernels on v_mfma_f32_16x16x32_f16 (the build) against the previous build (tools/ab/libs/chain32.so), same box; parity first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06s
python -m pytest tests/test_hip_parity.py tests/test_edge_batches.py tests/test_forward_only.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06s/parity.txt
tail -5 gpurun_out/r06s/parity.txt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
cp /tmp/lib_default.so tools/ab/libs/chain16.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in chain32 chain16; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if 'chain' in k})"
  done
done 2>&1 | tee gpurun_out/r06s/ab_chain16.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
