"""Where does the wall time of a training step go?  CPU-side (asynchronous) duration of each phase vs the synchronised total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
params = list(r.parameters())
opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.99), fused=True)
views = synthetic.synthetic_view(seed=1, device=dev)
sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(7))[:R].to(dev)
o, d, n, f, gt, m = [x[sel] for x in views]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import clip_per_parameter_
T = {}
def tick(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
def step(sync_each=False):
    t = time.perf_counter(); out = r(o, d, n, f); 
    if sync_each: torch.cuda.synchronize()
    tick("forward", t)
    t = time.perf_counter(); loss, _ = cn.compute_loss(out, gt, m)
    if sync_each: torch.cuda.synchronize()
    tick("loss", t)
    t = time.perf_counter()
    for p in params: p.grad = None
    loss.backward()
    if sync_each: torch.cuda.synchronize()
    tick("backward", t)
    t = time.perf_counter(); clip_per_parameter_(params)
    if sync_each: torch.cuda.synchronize()
    tick("clip", t)
    t = time.perf_counter(); opt.step()
    if sync_each: torch.cuda.synchronize()
    tick("adam", t)
for _ in range(3): step()
torch.cuda.synchronize()
for mode in (False, True):
    T.clear()
    t0 = time.perf_counter()
    for _ in range(10): step(mode)
    cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("sync_each=%s: wall %.2f ms/step, cpu-side %.2f ms/step; phases (ms/step): %s" % (mode, wall * 100, cpu * 100, {k: round(v * 100, 2) for k, v in T.items()}))
