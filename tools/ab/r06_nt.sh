#!/bin/bash
# cache-policy A/Bs (non-temporal loads / stores for rows touched once): alternates the variant libraries named in the `for v in ...` line (built by hand into tools/ab/libs/) three times; results: profiles/r06_ab_nt_*.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06nt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in base ntin; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().split('\n')[-1])
fam={e['kernel']:e['ms_per_step'] for e in b.get('kernel_breakdown',[])}
print('$v', b['value'], b['ms_per_step'], {k:fam[k] for k in fam if 'chain' in k})"
  done
done 2>&1 | tee gpurun_out/r06nt/ab_nts.txt
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
