"""Find a sensible torch thread count for the CPU baseline leg on the bench host (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import colorneus_oracle as O
ocfg = O.dtu_config()
P = {k: v.requires_grad_(True) for k, v in O.init_params(ocfg, seed=0, trained_like=True).items()}
R = 128
g = torch.Generator().manual_seed(1)
o = torch.randn(R, 3, generator=g); o = o / o.norm(dim=-1, keepdim=True) * 2.7
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.3 - o, dim=-1)
near, far = O.near_far_from_sphere(o, d)
gt = torch.rand(R, 3, generator=g); mask = (torch.rand(R, generator=g) < 0.7).float()
def step():
    out = O.render(P, ocfg, o, d, near, far, t_rand=torch.rand(R, 1))
    l, _ = O.compute_loss(out, gt, mask)
    for p in P.values(): p.grad = None
    l.backward()
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    step()
    t0 = time.perf_counter(); step(); step(); dt = (time.perf_counter() - t0) / 2
    print("threads %3d: %.2f s/iter  %.1f rays/s" % (nt, dt, R / dt), flush=True)
