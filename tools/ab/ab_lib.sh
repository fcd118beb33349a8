#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab/ab_lib.sh <other.so>   (the other build temporarily takes the place of the product library)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OTHER=$1
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
cp $R/color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
for rep in 1 2; do
  cp /tmp/lib_default.so $R/color-neus_amd/libcolorneus_hip.so
  python $R/bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default', b['value'], b['ms_per_step'])"
  cp $OTHER $R/color-neus_amd/libcolorneus_hip.so
  python $R/bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('other  ', b['value'], b['ms_per_step'])"
done
cp /tmp/lib_default.so $R/color-neus_amd/libcolorneus_hip.so
