#!/bin/bash
# Kernel-trace stats of the 512-ray step (the per-GPU share of C4) on the GPU box: bash tools/profile_small.sh r04   (through gpurun, repository root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-rXX}
OUT=$R/gpurun_out/${TAG}_prof512
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--rays 512 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > $OUT/stats.log 2>&1
cd $R
python tools/kernel_families.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_by_family.txt "# rocprofv3 --kernel-trace --stats over python3 bench.py --rays 512 --steps 20 --warmup 5 $ARGS (512 rays/step, 25 steps), instantiations merged by family"
python tools/launch_gaps.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) 0.5 > $OUT/launch_gaps.txt 2>&1
rm -rf $OUT/stats
cat $OUT/kernel_stats_by_family.txt | head -30; cat $OUT/launch_gaps.txt | head -4
