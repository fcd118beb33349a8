#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
python -m pytest tests/test_graph_step.py tests/test_clip_adam.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06c/graph_tests.txt
cat gpurun_out/r06c/graph_tests.txt
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
for rep in 1 2; do
  for v in graph nograph; do
    F=""; [ $v = nograph ] && F="--no-graph"
    python bench.py $ARGS $F 2>gpurun_out/r06c/err_$v.txt | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', b['value'], b['ms_per_step'], b['small_batch']['rays_per_step_per_gpu'], b['step_ms']['median'])"
  done
done 2>&1 | tee gpurun_out/r06c/ab_graph.txt
tail -5 gpurun_out/r06c/err_graph.txt
