#!/bin/bash
# stream form of the layer kernel: transposed product + epilogue straight out of the accumulators (direct) against the LDS transposition of the result (tpose); parity first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06w
python -m pytest tests/test_hip_parity.py tests/test_edge_batches.py tests/test_forward_only.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06w/parity.txt
tail -2 gpurun_out/r06w/parity.txt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in tpose direct; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if k in ('layer_dw','layer_gemm_ws')})"
  done
done 2>&1 | tee gpurun_out/r06w/ab_direct.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
