#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repository root): kernel-trace stats, HBM traffic (two PMC passes), SQ counters.
#   bash tools/profile_round.sh r02
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-rXX}
OUT=$R/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/sq -o q --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/sq.log 2>&1
cd $R
python tools/kernel_families.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_by_family.txt "# rocprofv3 --kernel-trace --stats over python3 bench.py --steps 5 --warmup 2 $ARGS (4096 rays/step, 7 steps), instantiations merged by family"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_hbm_traffic.json
python tools/pmc_summary.py $OUT/sq $OUT/pmc_sq_summary.txt > /dev/null
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
ls -la $OUT
