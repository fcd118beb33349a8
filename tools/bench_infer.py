"""Inference (validate_image-style, NeuS_Trainer.py:216-277) throughput: forward only, with and without early-termination compaction."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
views = synthetic.synthetic_view(seed=1, device=dev)
sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(7))[:R].to(dev)   # random pixels of the view
o, d, n, f, gt, m = [x[sel] for x in views]
res = {}
with torch.no_grad():
    ref = r(o, d, n, f, perturb_overwrite=0)
    for eps in (0.0, 1e-4, 1e-3):
        for _ in range(2): out = r(o, d, n, f, perturb_overwrite=0, prune_eps=eps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = r(o, d, n, f, perturb_overwrite=0, prune_eps=eps)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        kept = float((out["weights"] >= eps).float().mean()) if eps > 0 else 1.0
        res["eps=%g" % eps] = {"rays_per_s": round(R / dt, 1), "ms": round(dt * 1e3, 2), "kept_fraction": round(kept, 3),
                               "max_abs_color_diff": float((out["color_fine"] - ref["color_fine"]).abs().max())}
print(json.dumps({"config": "forward only, %d rays x 128 samples, DTU renderer block" % R, **res}))
