#!/bin/bash
# launch gaps at 512 / 4096 rays per step (run through gpurun from the repository root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for n in 512 4096; do
  rm -rf /tmp/gp$n
  rocprofv3 --kernel-trace -d /tmp/gp$n -o t --output-format csv -- python3 $R/bench.py --rays $n --steps 30 --warmup 5 $ARGS > $R/gpurun_out/gaps_$n.log 2>&1
  python3 $R/tools/launch_gaps.py $(find /tmp/gp$n -name "*kernel_trace.csv" | head -1) 0.5 > $R/gpurun_out/gaps_$n.txt 2>&1
  tail -1 $R/gpurun_out/gaps_$n.log | cut -c1-200
  cat $R/gpurun_out/gaps_$n.txt
done
