#!/bin/bash
# EXPERIMENT: upper bound of what pre-split staging could buy -- layer_dw whose staging waves only copy bytes (wrong results) against the product
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-optim --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in cur exp_copystage; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); kb={k['kernel']:k['ms_per_step'] for k in b['kernel_breakdown']}; print('$v', b['value'], b['ms_per_step'], 'layer_dw', kb.get('layer_dw'))"
  done
done 2>&1 | tee gpurun_out/r06_copystage.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
