#!/bin/bash
# review item 5 evidence: the forward-only call with the chain-fused gradient chain (CNR_CHAIN_GRAD=1) against the per-layer launches it replaces
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/r06q; mkdir -p $OUT
for rep in 1 2; do
  echo "per-layer gradient chain (default):"; python3 tools/bench_infer.py 8192 2>/dev/null | tail -1
  echo "chain-fused gradient chain (CNR_CHAIN_GRAD=1):"; CNR_CHAIN_GRAD=1 python3 tools/bench_infer.py 8192 2>/dev/null | tail -1
done > $OUT/infer_chain_grad_ab.txt 2>&1
cat $OUT/infer_chain_grad_ab.txt
cd /tmp && export TMPDIR=/tmp
export CNR_CHAIN_GRAD=1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/tools/bench_infer.py 8192 > $OUT/stats.log 2>&1
unset CNR_CHAIN_GRAD
cd $R
python tools/kernel_families.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_by_family_inference_chain_grad.txt "# CNR_CHAIN_GRAD=1: rocprofv3 --kernel-trace --stats over python3 tools/bench_infer.py 8192 (same call mix as r06_kernel_stats_by_family_inference.txt), instantiations merged by family"
rm -rf $OUT/stats
head -12 $OUT/kernel_stats_by_family_inference_chain_grad.txt
