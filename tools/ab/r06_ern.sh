#!/bin/bash
# layer_dw product waves: two alternating side-input register sets (ern2) against one set refilled in place (ern1: s_waitcnt vmcnt(0) + register copies at the end of every tile); parity first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06z
python -m pytest tests/test_hip_parity.py tests/test_edge_batches.py tests/test_forward_only.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06z/parity.txt
tail -2 gpurun_out/r06z/parity.txt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in ern1 ern2; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if k in ('layer_dw','layer_gemm_ws')})"
  done
done 2>&1 | tee gpurun_out/r06z/ab_ern.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
