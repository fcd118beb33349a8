#!/bin/bash
# general form of the layer kernel: 32x32x16 (gen32) / 16x16x32 as committed (noswz) / 16x16x32 with hoisted fragment loads + uniform dead-lane test (gen16): per-kernel times by rocprofv3
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/r06u; mkdir -p $OUT
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
cd /tmp && export TMPDIR=/tmp
for v in gen32 noswz gen16; do
  cp $R/tools/ab/libs/$v.so $R/color-neus_amd/libcolorneus_hip.so
  rm -rf $OUT/stats_$v
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$v -o s --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only > $OUT/log_$v.txt 2>&1
  echo "== $v"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'layer_gemm_ws' in r['Name']: print(r['Name'][:75].replace('void cnr::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
" $(find $OUT/stats_$v -name "*kernel_stats.csv" | head -1)
  rm -rf $OUT/stats_$v
done 2>&1 | tee $OUT/gen_ab.txt
cp /tmp/lib_default.so $R/color-neus_amd/libcolorneus_hip.so
