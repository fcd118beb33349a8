#!/usr/bin/env python3
"""Board power and shader clock PER KERNEL FAMILY (round 5, review item 5: "the power trace demonstrating the cap binds -- sclk and W per
kernel family, not per run").

rocm-smi samples every 0.25 s cannot resolve the launches inside one 24 ms step, so each family is LOOPED on its own for a few seconds
through the public entry points that reach it, and sampled while it runs:

  backward        cnr_render_backward over one saved forward (layer_dw is 80 % of its kernel time; plus sweep0 / narrow / head / strip / finish)
  backward_nomfma the same with CNR_FDW_DBG=5 (the fused launches without their MFMAs: the memory side alone; WRONG results) -- child process
  forward_saving  cnr_render_forward (sampler value chains, saving SDF chain, gradient chain, colour + relight chain)
  forward_only    cnr_render_forward_only
  sdf_value       cnr_sdf_eval on 2 M points (the chain-fused value kernel: MFMA + softplus epilogue, 196 B / point of HBM traffic)
  copy            torch copy of 1 GiB (a pure memory stream, for the scale of the power axis)
  idle            nothing

Usage (GPU box, from the repository root):  python tools/family_power.py [--seconds 4] [--rays 4096] > gpurun_out/r05_power_by_family.txt
Each line: family, launches-per-second figure of merit, W (mean / max of the samples after the first second), sclk (mean MHz).
"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
    except Exception:
        return None, None
    w = re.search(r"Power \(W\):\s*([0-9.]+)", t)
    c = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", t)
    return (float(w.group(1)) if w else None), (float(c.group(1)) if c else None)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.samples = []

    def run(self):
        t0 = time.time()
        while not self.stop:
            w, c = smi()
            self.samples.append((time.time() - t0, w, c))
            time.sleep(0.25)


def run_family(name, seconds, rays):
    import torch
    import color_neus_amd as cn
    from color_neus_amd import synthetic
    dev = torch.device("cuda:0")
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
    torch.manual_seed(0)
    libpath = os.environ.get("CNR_LIB") or None   # backward_nomfma: the tuning build (make hip-tuning), the product has no ablation words
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg, library=libpath)).to(dev)
    views = synthetic.synthetic_view(seed=1, device=dev)
    sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(7))[:rays].to(dev)
    o, d, n, f, gt, m = [x[sel] for x in views]
    unit = "steps/s"
    if name.startswith("backward"):
        def body():
            out = r(o, d, n, f, perturb_overwrite=0)
            loss, _ = cn.compute_loss_fused(out, gt, m, library=cn.load_library(libpath))
            # the same graph backward again and again: only cnr_render_backward (+ the loss backward launch) repeats
            for _ in range(8):
                for p in r.parameters():
                    p.grad = None
                loss.backward(retain_graph=True)
            return 8
        unit = "backward passes/s"
    elif name == "forward_saving":
        def body():
            with torch.no_grad():
                for _ in range(4):
                    r(o, d, n, f, perturb_overwrite=0, forward_only=False)
            return 4
        unit = "forward passes/s"
    elif name == "forward_only":
        def body():
            with torch.no_grad():
                for _ in range(4):
                    r(o, d, n, f, perturb_overwrite=0)
            return 4
        unit = "forward passes/s"
    elif name == "sdf_value":
        pts = torch.rand(1 << 21, 3, device=dev) * 2 - 1

        def body():
            r.sdf(pts)
            return 1
        unit = "calls of 2 M points/s"
    elif name == "copy":
        a = torch.empty(1 << 28, device=dev)
        b = torch.empty_like(a)

        def body():
            for _ in range(8):
                b.copy_(a)
            return 8
        unit = "GiB copies/s"
    elif name == "idle":
        def body():
            time.sleep(0.2)
            return 0
    else:
        raise SystemExit("unknown family " + name)
    body()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.time()
    n_done = 0
    while time.time() - t0 < seconds:
        n_done += body()
        torch.cuda.synchronize()
    dt = time.time() - t0
    s.stop = True
    s.join(timeout=3)
    ws = [w for t, w, c in s.samples if t > 1.0 and w is not None]
    cs = [c for t, w, c in s.samples if t > 1.0 and c is not None]
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    print("%-16s %8.2f %-24s W mean %7.1f max %7.1f   sclk mean %6.0f MHz   (%d samples, %d rays)" %
          (name, n_done / dt, unit, mean(ws), max(ws) if ws else float("nan"), mean(cs), len(ws), rays), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--family", default=None)
    a = ap.parse_args()
    if a.family:
        run_family(a.family, a.seconds, a.rays)
        return
    w, c = smi()
    print("# rocm-smi every 0.25 s while ONE kernel family loops (tools/family_power.py); samples of the first second dropped; idle now: %s W, sclk %s MHz" % (w, c))
    cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
    mcap = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", cap)
    print("# power cap: %s W" % (mcap.group(1) if mcap else "?"))
    # every family in a child process of its own (the library reads its debugging switches once per process)
    for fam, env in (("idle", {}), ("copy", {}), ("sdf_value", {}), ("forward_only", {}), ("forward_saving", {}), ("backward", {}),
                     ("backward_nomfma", {"CNR_FDW_DBG": "5", "CNR_LIB": os.path.join(ROOT, "tools", "_build", "libcolorneus_hip_tuning.so")})):
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--family", fam, "--seconds", str(a.seconds), "--rays", str(a.rays)], env=e)


if __name__ == "__main__":
    main()
