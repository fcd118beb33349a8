"""Copy a profile round (gpurun_out/<tag>_prof, <tag>_bench.json, <tag>_bench_eval.json, param_grad_error_table.txt) into profiles/ under the
round's names, stamped with the build (git HEAD).  Usage: python tools/install_profiles.py r02d r02"""
import json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
H = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=root).decode().strip()
src, dst = os.path.join(root, "gpurun_out", tag + "_prof") + "/", os.path.join(root, "profiles") + "/"
d = json.load(open(src + "pmc_hbm_traffic.json"))
steps = 3   # bench.py --steps 2 --warmup 1
d["build"], d["steps_profiled"] = H, steps
d["hbm_bytes_per_step_all_kernels"] = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in d["kernels"].items()
                                                if k not in ("layer_gemm_ws_general", "layer_gemm_ws_stream")) / steps)   # (layer_gemm_ws = both forms)
json.dump(d, open(dst + rnd + "_pmc_hbm_traffic_4096rays.json", "w"), indent=1)
open(dst + rnd + "_kernel_stats_by_family.txt", "w").write("# build %s\n" % H + open(src + "kernel_stats_by_family.txt").read())
shutil.copy(src + "kernel_stats.csv", dst + rnd + "_kernel_stats_4096rays.csv")
hdr = ("# build %s; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES "
       "SQ_INSTS_VALU over python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 (percentages of SQ_WAVE_CYCLES)\n" % H)
open(dst + rnd + "_pmc_sq_summary.txt", "w").write(hdr + open(src + "pmc_sq_summary.txt").read())
b = json.loads(open(os.path.join(root, "gpurun_out", tag + "_bench.json")).read().strip().split("\n")[-1])
b["build"] = H
if b.get("roofline"):
    b["roofline"]["traffic"] = d["kernels"].get(b["roofline"].get("kernel_name", "layer_gemm_ws"), {}).get("hbm_bytes_per_launch")   # the value bench.py reads from now on
json.dump(b, open(dst + rnd + "_bench_4096rays.json", "w"), indent=1)
t = open(os.path.join(root, "gpurun_out", "param_grad_error_table.txt")).read()
open(dst + rnd + "_param_grad_error_table.txt", "w").write("# build %s; written by tests/test_hip_parity.py::test_param_grad_error_table on an MI355X (HIP path vs the reference's "
                                                            "float64 runs, own scale per tensor)\n" % H + t)
ev = os.path.join(root, "gpurun_out", tag + "_bench_eval.json")
if os.path.exists(ev):
    e = json.loads([l for l in open(ev).read().strip().split("\n") if l.startswith("{")][-1])
    e["build"] = H
    json.dump(e, open(dst + rnd + "_bench_eval.json", "w"), indent=1)
print("build", H, "GB/step", d["hbm_bytes_per_step_all_kernels"] / 1e9)
for k in ("value", "ms_per_step", "small_batch", "kernel_ms_per_step", "launches_per_step"):
    print(k, b.get(k))
print(b["torch_gpu_baseline"]["value"], b["torch_gpu_baseline"]["speedup_at_4096"], b["torch_gpu_baseline"]["speedup_at_equal_batch"])
print({k: b["roofline"][k] for k in ("achieved", "frac", "mfma_frac", "avg_launch_ms", "traffic", "algorithmic_bytes_per_launch")})
