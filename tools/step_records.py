"""Per-launch records of one training step (library timing scopes: name, shape, ms, algorithmic bytes).  Usage on the GPU box:
  python tools/step_records.py [rays=4096] [min_ms=0.05] [n_importance=64]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
dev = torch.device("cuda:0")
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 4096   # rays per step
THR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05   # print records above this many ms
NI = int(sys.argv[3]) if len(sys.argv) > 3 else 64   # importance samples per ray (0: the C2 form, 64 coarse samples only)
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0, n_importance=NI)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
lib = cn.load_library()
o, d, near, far, gt, mask = [x[:NR] for x in synthetic.synthetic_view(seed=1, device=dev)]
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
    if ms > THR: print("%-18s nt=%3d N=%4d K=%4d pairs=%d  %.3f ms  %.2f GB  %.2f TB/s" % (name, nt, N, K, pairs, ms, nbytes/1e9, nbytes/ms/1e9 if ms else 0))
import collections
agg = collections.OrderedDict()
for name, kind, nt, P, N, K, pairs, ms, nbytes in recs:
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
print("--- by name (rays %d): total %.3f ms" % (NR, sum(v[1] for v in agg.values())))
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%-18s x%3d  %.3f ms" % (k, n, ms))
