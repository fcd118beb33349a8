"""debug: one 65536-ray forward-only call against eight 8192-ray calls on the same rays"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import color_neus_amd as cn
from color_neus_amd import synthetic, rays as raygen
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
lib = cn.load_library()
c2w, focal, image, mask = synthetic.synthetic_camera(800, 800, seed=1, device=dev)
vo, vd = raygen.get_rays_at(c2w[0], focal, 800, 800, normalize=True, library=lib)
vo, vd = vo.reshape(-1, 3), vd.reshape(-1, 3)
vn, vf = raygen.near_far_from_sphere(vo, vd)
for N in (16384, 32768, 65536):
    o, d, n, f = vo[:N], vd[:N], vn[:N], vf[:N]
    for kw in (dict(), dict(prune_eps=1e-4)):
        with torch.no_grad():
            big = r(o, d, n, f, perturb_overwrite=0, **kw)
            parts = [r(o[a:a + 8192], d[a:a + 8192], n[a:a + 8192], f[a:a + 8192], perturb_overwrite=0, **kw) for a in range(0, N, 8192)]
        torch.cuda.synchronize()
        for k in ("color_fine", "depth", "weights", "gradients", "z_vals", "delta_relight", "cdf_fine"):
            small = torch.cat([p[k].reshape(8192, -1) for p in parts], 0)
            b = big[k].reshape(N, -1)
            diff = (b != small).any(dim=1)
            nd = int(diff.sum())
            if nd:
                idx = torch.nonzero(diff).reshape(-1)
                print(N, kw, k, "rays differing:", nd, "first", int(idx[0]), "last", int(idx[-1]), "max abs", float((b - small).abs().max()))
            else:
                print(N, kw, k, "equal")
