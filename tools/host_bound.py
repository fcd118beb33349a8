#!/usr/bin/env python3
"""Is the small-batch step host-bound?  Per step at R rays: the host time to ENQUEUE the step (no synchronisation inside the loop) against the
GPU time between the step's two stream events.  If the two are close the GPU waits for the host, and the rate follows the host's speed.
Usage: python tools/host_bound.py [rays ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic, rays as raygen
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
lib = cn.load_library()
opt = cn.ClipAdam(r._ordered_params(), lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=1.0, library=lib)
c2w, focal, image, mask = synthetic.synthetic_camera(seed=1, device=dev)
perm = torch.randperm(640000, generator=torch.Generator().manual_seed(7)).to(dev)
params = list(r.parameters())
def step(i, R):
    idx = perm[(i * R) % (640000 - R):(i * R) % (640000 - R) + R]
    o, d, rgb, msel, near, far = raygen._generate(lib, idx, R, c2w, focal, 800, 800, True, False, image=image, mask=mask, origin=None, radius=1.0, want_nearfar=True)
    out = r(o, d, near, far)
    loss, _ = cn.compute_loss_fused(out, rgb, msel, library=lib)
    for p in params: p.grad = None
    loss.backward()
    opt.step()
for R in [int(a) for a in sys.argv[1:]] or [512, 1024, 4096]:
    for i in range(10): step(i, R)
    torch.cuda.synchronize()
    n = 60
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    t_all0 = time.perf_counter()
    for i in range(n):
        ev[i].record()
        t0 = time.perf_counter()
        step(10 + i, R)
        host.append(time.perf_counter() - t0)
    ev[n].record()
    t_enq = time.perf_counter() - t_all0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all0
    g = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    h = sorted(host)
    print("R=%5d  host enqueue ms/step: median %.3f p90 %.3f | GPU ms/step (events): median %.3f | wall %.3f ms/step, enqueue loop done after %.1f %% of the wall time" %
          (R, h[n // 2] * 1e3, h[int(0.9 * n)] * 1e3, g[n // 2], t_all / n * 1e3, 100 * t_enq / t_all))
