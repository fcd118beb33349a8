#!/bin/bash
# remaining hot-path shuffles (sweep0_dw, narrow_bwd, strip_bwd's halving exchange) as DPP moves / lane swaps (swap2) against the build before (swap1); parity first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06x
python -m pytest tests/test_hip_parity.py tests/test_edge_batches.py tests/test_forward_only.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06x/parity.txt
tail -2 gpurun_out/r06x/parity.txt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in swap1 swap2; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if k in ('sweep0_dw','narrow_bwd','strip_bwd','head_bwd')})"
  done
done 2>&1 | tee gpurun_out/r06x/ab_swap2.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
