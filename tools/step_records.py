import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
lib = cn.load_library()
o, d, near, far, gt, mask = [x[:4096] for x in synthetic.synthetic_view(seed=1, device=dev)]
def step():
    out = r(o, d, near, far)
    loss, _ = cn.compute_loss_fused(out, gt, mask, library=lib)
    for p in r.parameters(): p.grad = None
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
lib.timing_enable(True); step(); torch.cuda.synchronize()
recs = lib.timing_collect(); lib.timing_enable(False)
for name, kind, nt, P, N, K, pairs, ms, nbytes in recs:
    if ms > 0.05: print("%-18s nt=%3d N=%4d K=%4d pairs=%d  %.3f ms  %.2f GB  %.2f TB/s" % (name, nt, N, K, pairs, ms, nbytes/1e9, nbytes/ms/1e9 if ms else 0))
