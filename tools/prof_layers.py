"""Per-launch timing of one training step (HIP events via the library's timing hooks).  Usage: python tools/prof_layers.py [rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg, library=os.environ.get('CNR_LIB'))).to(dev)
views = synthetic.synthetic_view(seed=1, device=dev)
sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(7))[:R].to(dev)
o, d, n, f, gt, m = [x[sel] for x in views]
lib = cn.load_library(os.environ.get('CNR_LIB'))
def step():
    out = r(o, d, n, f)
    loss, _ = cn.compute_loss(out, gt, m)
    for p in r.parameters(): p.grad = None
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
lib.timing_enable(True); step(); torch.cuda.synchronize(); recs = lib.timing_collect(); lib.timing_enable(False)
tot = sum(x[7] for x in recs)
print("total kernel ms %.2f over %d launches" % (tot, len(recs)))
if os.environ.get("CNR_BRIEF"):
    recs = []
for i, (name, kind, nt, P, N, K, pairs, ms, nb) in enumerate(recs):
    fl = 2.0 * P * N * K * max(pairs, 1) if kind != 2 else 0
    print("%3d %-16s nt=%-5d P=%-8d N=%-4d K=%-4d pairs=%d  %8.3f ms  %6.1f TF/s %7.1f GB/s" % (i, name, nt, P, N, K, pairs, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, nb / (ms * 1e-3) / 1e9 if ms > 0 else 0))
