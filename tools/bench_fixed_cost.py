"""Per-launch times of the layer kernels at small point counts (fixed cost of a launch): python tools/bench_fixed_cost.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rays in (32, 128, 512, 2048):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rays", str(rays), "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
                          "--no-torch-gpu-baseline", "--no-small-batch"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
    d = json.loads(out)
    kb = {k["kernel"]: k for k in d["kernel_breakdown"]}
    print("rays %5d  step %.3f ms  kernels %.3f ms  launches %d | " % (rays, d["ms_per_step"], d["kernel_ms_per_step"], d["launches_per_step"]) +
          "  ".join("%s %.1f us x%d" % (n, 1e3 * kb[n]["ms_per_step"] / kb[n]["launches"], kb[n]["launches"]) for n in ("layer_gemm_ws", "dw_gemm_hx", "layer_gemm", "chain_sdf_value", "finish_weight", "clip_adam") if n in kb))
