#!/bin/bash
# same-box comparison of several values of one environment variable: bash tools/ab/ab_vals.sh CNR_FDW_DEEP 0 1 2 3   (through gpurun, repository root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
VAR=$1; shift
ARGS="--rays ${RAYS:-4096} --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v python $R/bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$VAR=$v', b['value'], b['ms_per_step'])"
  done
done
