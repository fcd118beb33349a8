"""Value-chain tile shapes side by side (CNR_CHAIN_SHAPE is read once per process: one child per shape): time per call on 2 M points, board power and clock
while it loops, and a checksum of the output (shapes must agree to round-off).  Usage (GPU box): python tools/shape_probe.py 41 22 12
(the <4, 2> experiment of DESIGN.md 4.9 was `else if (shape == 42) launch_sdf_value_chain<4, 2>(c, s);` in be_sdf_value_chain with launch bounds (256, 1); not kept)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch, time
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import family_power as F
    import color_neus_amd as cn
    from color_neus_amd import synthetic
    dev = torch.device("cuda:0")
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
    torch.manual_seed(0)
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
    pts = torch.rand(1 << 21, 3, generator=torch.Generator().manual_seed(1)).to(dev) * 2 - 1
    ref = r.sdf(pts).double()
    torch.cuda.synchronize()
    s = F.Sampler(); s.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(10): r.sdf(pts)
        torch.cuda.synchronize(); n += 10
    dt = time.time() - t0
    s.stop = True; s.join(timeout=3)
    ws = [w for t, w, c in s.samples if t > 1.0 and w]; cs = [c for t, w, c in s.samples if t > 1.0 and c]
    print("shape %s: %.3f ms per call of 2 M points   W mean %.0f   sclk %.0f MHz   sum %.9f  abs-sum %.9f" %
          (os.environ.get("CNR_CHAIN_SHAPE", "default"), dt / n * 1e3, sum(ws) / max(len(ws), 1), sum(cs) / max(len(cs), 1), float(ref.sum()), float(ref.abs().sum())), flush=True)
else:
    for sh in sys.argv[1:] or ["41", "22", "12"]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, CNR_CHAIN_SHAPE=sh))
