#!/bin/bash
# round 6, first GPU session: parity of the tr_b16 build, same-box A/B against the two-byte gathers, the energy ledger
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "g2 or fused or fallback or every_gradient" 2>&1 | tail -5 > gpurun_out/r06a/parity.txt
cp color-neus_amd/libcolorneus_hip.so /tmp/lib_default.so
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2 3; do
  for v in tr0 tr1; do
    cp tools/ab/libs/$v.so color-neus_amd/libcolorneus_hip.so
    python bench.py $ARGS 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', b['value'], b['ms_per_step'])"
  done
done > gpurun_out/r06a/ab_tr.txt 2>&1
cp /tmp/lib_default.so color-neus_amd/libcolorneus_hip.so
python tools/energy_ledger.py --seconds 4 > gpurun_out/r06a/energy_ledger.txt 2> gpurun_out/r06a/energy_ledger.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-c5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
cat gpurun_out/r06a/parity.txt gpurun_out/r06a/ab_tr.txt; head -30 gpurun_out/r06a/energy_ledger.txt
