"""BASELINE config 5: evaluation.py mesh extraction at -rr 512 = dense SDF lattice (extract_fields, NeuS.py:14-28) + per-vertex
colour (extract_color, NeuS.py:44-64) on one MI355X.  Prints one JSON line.  (Marching cubes itself is third-party CPU code, out of scope.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
u = r.extract_fields([-1.01] * 3, [1.01] * 3, dev, 64)   # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
u = r.extract_fields([-1.01] * 3, [1.01] * 3, dev, res)
torch.cuda.synchronize(); t1 = time.perf_counter()
u_host = u.cpu(); t2 = time.perf_counter()
mv, mt = r.marching_cubes(u, [-1.01] * 3, [1.01] * 3, 0.0)   # device iso-surface on the resident lattice (replaces the D2H + CPU mcubes)
torch.cuda.synchronize(); t2b = time.perf_counter()
g = torch.Generator().manual_seed(3)
v = torch.randn(nv, 3, generator=g); v = (v / v.norm(dim=-1, keepdim=True) * (0.5 + 0.02 * torch.randn(nv, 1, generator=g))).numpy()
r.extract_color(v[:1000], dev)
torch.cuda.synchronize(); t3 = time.perf_counter()
rgb = r.extract_color(v, dev)
t4 = time.perf_counter()
n = res ** 3
flop = 2.0 * 524544 * n
print(json.dumps({"config": "C5 grid %d^3 + %d vertex colours" % (res, nv), "grid_s": round(t1 - t0, 3), "grid_Mpts_per_s": round(n / (t1 - t0) / 1e6, 1),
                  "grid_TFLOPs": round(flop / (t1 - t0) / 1e12, 1), "d2h_s": round(t2 - t1, 3), "marching_cubes_s": round(t2b - t2, 3), "mesh_vertices": int(mv.shape[0]), "mesh_triangles": int(mt.shape[0]), "vertex_colour_s": round(t4 - t3, 3),
                  "vertex_Mpts_per_s": round(nv / (t4 - t3) / 1e6, 2), "sdf_inside_fraction": float((u_host > 0).float().mean())}))
