#!/bin/bash
# experiment: the weight-gradient product of layer_dw with one or both cross terms of the split-f16 product dropped (FD_DW_DROP = 1: no hi x lo, 2: no lo x hi, 3: hi x hi only):
# do the parity gates hold, and what does it buy?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06dw
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only --no-roofline"
for v in 0 1 2 3; do
  cp tools/ab/libs/dwdrop$v.so color-neus_amd/libcolorneus_hip.so
  echo "== FD_DW_DROP=$v"
  python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench', b['value'], b['ms_per_step'])"
  if [ $v != 0 ]; then
    python -m pytest tests/test_hip_parity.py -q -m gpu -k "g2_render_core" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-150 | tail -12
    python -m pytest tests/test_full_size_oracle.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|AssertionError" | cut -c1-300 | tail -6
  fi
done 2>&1 | tee gpurun_out/r06dw/dwdrop.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
