#!/bin/bash
# same-box A/B of one environment switch: bash tools/ab/ab_env.sh CNR_NO_SWEEP0 [rays]   (run through gpurun from the repository root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
SW=$1; RAYS=${2:-4096}
ARGS="--rays $RAYS --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2; do
  python $R/bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default      ', b['value'], b['ms_per_step'])"
  env $SW=1 python $R/bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$SW=1', b['value'], b['ms_per_step'])"
done
