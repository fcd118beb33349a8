#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
python -m pytest tests/test_hip_parity.py tests/test_edge_batches.py -x -q -m gpu -k "g2 or fused or fallback or every_gradient or ragged or edge" 2>&1 | tail -5 > gpurun_out/r06n/parity.txt
cat gpurun_out/r06n/parity.txt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in mfma16 ws16; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', b['value'], b['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06n/ab_ws16.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
