#!/bin/bash
# Board power / clocks sampled with rocm-smi while bench.py runs (run through gpurun from the repository root); writes gpurun_out/<tag>_power_samples.txt
#   bash tools/power_samples.sh r03
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-rXX}
OUT=$R/gpurun_out/${TAG}_power_samples.txt
mkdir -p $R/gpurun_out
{
  echo "# rocm-smi samples (every 0.5 s) around: python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-torch-gpu-baseline --no-inference --no-c5 --no-small-batch --no-roofline"
  echo "# idle sample first:"
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Max" | sed 's/^/idle: /'
} > $OUT
python $R/bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-torch-gpu-baseline --no-inference --no-c5 --no-small-batch --no-roofline > $R/gpurun_out/${TAG}_power_bench.json 2>/dev/null &
BP=$!
i=0
while kill -0 $BP 2>/dev/null; do
  echo "sample $i (0.5 s apart) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -E 's/.*GPU\[0\]\s*:\s*//' | tr '\n' ';')" >> $OUT
  sleep 0.5; i=$((i+1))
done
wait $BP
echo "# bench line of this run:" >> $OUT
python -c "import json;d=json.load(open('$R/gpurun_out/${TAG}_power_bench.json'));print('# value', d['value'], 'rays/s, ms_per_step', d['ms_per_step'])" >> $OUT
tail -5 $OUT
