"""Measured errors behind the gates of tests/test_all_output_grads.py, the N_OUTSIDE test and the 4096-ray additivity test (GPU box):
prints, per tensor, the HIP error against float64 next to the float32 oracle's own error, so that the gates can be set from measurements.
Usage: python tools/gate_probe.py [all|outgrads|outside|additivity]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import _golden as G
import _native as N
from oracle import colorneus_oracle as O

DEV = "cuda:0" if torch.cuda.is_available() else "cpu"
LIB = None if torch.cuda.is_available() else N.EMU_LIB


def outgrads(name, cos_anneal=0.3, bg=(0.2, 0.5, 0.7)):
    import test_all_output_grads as T
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    g = torch.Generator().manual_seed(123)
    z = torch.from_numpy(fx["jit:z_vals"])
    o, d = torch.from_numpy(fx["rays_o"]), torch.from_numpy(fx["rays_d"])
    near, far = torch.from_numpy(fx["jit:near"]), torch.from_numpy(fx["jit:far"])
    bgt = torch.tensor(bg)
    ref = {}
    coefs = None
    for dt in (torch.float64, torch.float32):
        Pd = {k: v.to(dt).clone().requires_grad_(True) for k, v in P.items()}
        od, dd = o.to(dt).clone().requires_grad_(True), d.to(dt).clone().requires_grad_(True)
        out_o = O.render(Pd, ocfg, od, dd, near.to(dt), far.to(dt), z_vals=z.to(dt), cos_anneal_ratio=cos_anneal, background_rgb=bgt.to(dt))
        if coefs is None:
            coefs = {k: torch.randn(out_o[k].shape, generator=g, dtype=torch.float64) for k in T.KEYS if k in out_o}
            coefs["weights"] *= 3.0
            coefs["gradient_error"] = coefs["gradient_error"] * 5.0
        L = sum((out_o[k] * coefs[k].to(dt)).sum() for k in coefs)
        L.backward()
        gr = {k: v.grad.detach() for k, v in Pd.items()}
        gr["rays_o"], gr["rays_d"] = od.grad, dd.grad
        ref[dt] = ({k: v.detach() for k, v in out_o.items() if torch.is_tensor(v)}, gr)
    r = N.make_renderer(ocfg, P, LIB, DEV)
    on, dn = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    out_n = r(on, dn, near.to(DEV), far.to(DEV), z_vals=z.to(DEV), cos_anneal_ratio=cos_anneal, background_rgb=bgt)
    L_n = sum((out_n[k] * coefs[k].float().to(DEV).reshape(out_n[k].shape)).sum() for k in coefs)
    L_n.backward()
    print(f"## all-output cotangents: {name} on {DEV}")
    o64, g64 = ref[torch.float64]
    o32, g32 = ref[torch.float32]
    for k in coefs:
        print("out %-20s hip %.2e   f32-oracle %.2e" % (k, G.relerr(out_n[k].detach().cpu().reshape(o64[k].shape), o64[k]), G.relerr(o32[k].double(), o64[k])))
    got = {(k[len("renderer."):] if k.startswith("renderer.") else k): p.grad for k, p in r.named_parameters()}
    got["rays_o"], got["rays_d"] = on.grad, dn.grad
    print("%-44s %8s %10s %10s %10s %10s" % ("gradient", "numel", "hip_max", "hip_bulk1%", "f32_max", "f32_bulk1%"))
    for k, r64 in g64.items():
        r64 = r64.double().reshape(-1)
        den = max(float(r64.abs().max()), 1e-300)
        e = (got[k].detach().cpu().double().reshape(-1) - r64).abs() / den
        e32 = (g32[k].double().reshape(-1) - r64).abs() / den
        allowed = G._allowed(e.numel(), True)
        bulk = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        bulk32 = float(torch.sort(e32).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        print("%-44s %8d %10.2e %10.2e %10.2e %10.2e" % (k, e.numel(), float(e.max()), bulk, float(e32.max()), bulk32))


def outside():
    for name in ("tiny_outside", "tiny_neus_outside"):
        for tag in ("det", "jit"):
            fx, r, out, loss, grads, o, d = N.run_native(name, tag, LIB, DEV, fixed_z=True)
            rows = G.param_grad_table(fx, tag, grads, strict=True)
            print(G.format_grad_table(f"{name}/{tag} (strict columns)", rows))


def additivity():
    import color_neus_amd as cn
    from color_neus_amd import synthetic
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
    torch.manual_seed(0)
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(DEV)
    views = synthetic.synthetic_view(seed=1, device=DEV)
    sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(11))[:4096].to(DEV)
    o, d, near, far, gt, mask = [x[sel] for x in views]
    t_rand = torch.rand(4096, 1, generator=torch.Generator().manual_seed(5))

    def run(idx):
        orig = torch.rand
        try:
            torch.rand = lambda *a, **k: t_rand[idx.cpu()].clone()
            out = r(o[idx], d[idx], near[idx], far[idx])
        finally:
            torch.rand = orig
        obj = ((out["color_fine"] - gt[idx]) ** 2).sum() + 0.1 * out["weight_sum"].sum() + 0.01 * (out["gradients"] ** 2).sum() \
            + 0.05 * out["delta_relight"].sum()
        for p in r.parameters():
            p.grad = None
        obj.backward()
        return out, {k: p.grad.clone() for k, p in r.named_parameters()}
    allr = torch.arange(4096, device=DEV)
    _, g_all = run(allr)
    _, g_a = run(allr[:2048])
    _, g_b = run(allr[2048:])
    print("## additivity at 4096 rays: |g_all - (g_a + g_b)|_max / |g_all|_max")
    for k, g in g_all.items():
        print("%-44s %.2e" % (k, float((g - (g_a[k] + g_b[k])).abs().max()) / float(g.abs().max())))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "outgrads"):
        for n in ("tiny_sharp", "tiny_neus_sharp", "dtu_sharp"):
            outgrads(n)
    if what in ("all", "outside"):
        outside()
    if what in ("all", "additivity") and torch.cuda.is_available():
        additivity()
