#!/bin/bash
# Kernel-trace stats of the forward-only inference call (8192 rays, without / with compaction) on the GPU box: bash tools/profile_infer.sh r05   (through gpurun, repository root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-rXX}
OUT=$R/gpurun_out/${TAG}_profinfer
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/tools/bench_infer.py 8192 > $OUT/stats.log 2>&1
cd $R
python tools/kernel_families.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_by_family.txt "# rocprofv3 --kernel-trace --stats over python3 tools/bench_infer.py 8192 (forward-only entry point: 1 + 7 calls without compaction, 7 each at prune_eps 1e-4 and 1e-3; 8192 rays x 128 samples per call), instantiations merged by family"
rm -rf $OUT/stats
cat $OUT/kernel_stats_by_family.txt | head -30; tail -2 $OUT/stats.log | cut -c1-600
