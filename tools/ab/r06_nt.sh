#!/bin/bash
# epilogue side inputs as non-temporal loads (nt) against plain loads (base)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06nt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in base ntS; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if k in ('layer_dw','sweep0_dw')})"
  done
done 2>&1 | tee gpurun_out/r06nt/ab_nts.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
