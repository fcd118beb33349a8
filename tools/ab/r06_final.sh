#!/bin/bash
# round 6: bench line of the build + rocprofv3 summaries (kernel stats, HBM traffic, SQ counters; 512-ray step; inference call) + ledger
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err
tail -c 300 gpurun_out/r06h_bench.json
bash tools/profile_round.sh r06h > gpurun_out/r06h_profile.log 2>&1
bash tools/profile_small.sh r06h > gpurun_out/r06h_profile_small.log 2>&1
bash tools/profile_infer.sh r06h > gpurun_out/r06h_profile_infer.log 2>&1
python tools/energy_ledger.py --seconds 4 > gpurun_out/r06h_energy_ledger.txt 2> gpurun_out/r06h_energy_ledger.err
head -14 gpurun_out/r06h_energy_ledger.txt
