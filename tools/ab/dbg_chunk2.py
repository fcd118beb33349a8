import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import color_neus_amd as cn
from color_neus_amd import synthetic, rays as raygen, parallel
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
lib = cn.load_library()
c2w, focal, image, mask = synthetic.synthetic_camera(800, 800, seed=1, device=dev)
vo, vd = raygen.get_rays_at(c2w[0], focal, 800, 800, normalize=True, library=lib)
vo, vd = vo.reshape(-1, 3), vd.reshape(-1, 3)
vn, vf = raygen.near_far_from_sphere(vo, vd)
a = parallel.sharded_render_image(r, vo, vd, vn, vf, chunk=8192, perturb_overwrite=0)
b = parallel.sharded_render_image(r, vo, vd, vn, vf, chunk=65536, perturb_overwrite=0)
a2 = parallel.sharded_render_image(r, vo, vd, vn, vf, chunk=8192, perturb_overwrite=0)
torch.cuda.synchronize()
for k in a:
    for name, x, y in (("8192 vs 65536", a[k], b[k]), ("8192 vs 8192 again", a[k], a2[k])):
        diff = (x.reshape(x.shape[0], -1) != y.reshape(y.shape[0], -1)).any(dim=1)
        idx = torch.nonzero(diff).reshape(-1)
        print(k, name, "rays differing", int(diff.sum()), ("first %d last %d max abs %.3e" % (int(idx[0]), int(idx[-1]), float((x - y).abs().max()))) if len(idx) else "")
        if len(idx):
            print("   sample of indices:", idx[:8].tolist(), idx[-8:].tolist())
