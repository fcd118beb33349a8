"""The three library steps of the N_OUTSIDE > 0 path (color-neus_amd/background.py: OutsideZ, Background, CompositeBg) one by one, forward AND
backward for every input, against autograd of the float64 torch restatement (oracle/background_oracle.py, itself pinned to the reference's
goldens by tests/test_oracle_golden.py) under random cotangents.  The end-to-end fixtures (tiny_outside) cannot do this for d far / d z:
there the background sits behind an opaque surface and its gradients underflow.
CPU: emulation build; GPU (-m gpu): HIP build."""
import os

import pytest
import torch

import _golden as G
import _native as N
from oracle import background_oracle as BO


def _lib(library):
    import color_neus_amd as cn
    return cn.load_library(library)


def _rays(R, g):
    o = torch.randn(R, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 2.5
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.4 - o, dim=-1)
    return o, d


def _outside_z(library, dev, perturb):
    from color_neus_amd import background as B
    lib = _lib(library)
    g = torch.Generator().manual_seed(1)
    R, M, n_out, S = 37, 24, 8, 16
    far = (torch.rand(R, generator=g) * 2 + 2.5)
    z = torch.sort(torch.rand(R, M, generator=g) * 2 + 0.7, dim=-1).values
    z[3, 5] = z[3, 4]                                                  # a tie
    t = torch.rand(R, n_out, generator=g) if perturb else None
    coef = torch.randn(R, M + n_out, generator=g, dtype=torch.float64)
    # oracle (float64)
    f64, z64 = far.double().requires_grad_(True), z.double().requires_grad_(True)
    zz = torch.linspace(1e-3, 1.0 - 1.0 / (n_out + 1.0), n_out).double()
    if perturb:
        mids = 0.5 * (zz[1:] + zz[:-1])
        upper, lower = torch.cat([mids, zz[-1:]]), torch.cat([zz[:1], mids])
        zz = lower[None] + (upper - lower)[None] * t.double()
    zo = f64[:, None] / torch.flip(zz, dims=[-1]) + 1.0 / S
    zf = torch.sort(torch.cat([z64, zo.expand(R, n_out)], -1), -1).values
    (zf * coef).sum().backward()
    # native
    fn, zn = far.to(dev).requires_grad_(True), z.to(dev).requires_grad_(True)
    z_feed, src = B.OutsideZ.apply(lib, fn, t.to(dev) if perturb else None, zn, S, n_out)
    (z_feed * coef.float().to(dev)).sum().backward()
    assert float((z_feed.detach().cpu().double() - zf.detach()).abs().max()) < 2e-6 * float(zf.detach().abs().max())
    assert bool((z_feed[:, 1:] >= z_feed[:, :-1]).all())
    assert G.relerr(fn.grad.cpu(), f64.grad) < 1e-5
    assert G.relerr(zn.grad.cpu(), z64.grad) < 1e-5


def _background(library, dev, small):
    """small: a 4 x 64 network with a skip on 80 points -- about 2e4 ReLU decisions, none of them within round-off of the kink, so EVERY gradient
    entry is held to 2e-4 of its tensor's scale.  Full size (the NeRF() defaults, 8 x 256 on 399 points = 8e5 decisions): one or two units sit
    within float32 round-off of zero and whichever side an implementation rounds them to moves that point's whole contribution to the layers
    below (see tests/_golden.py check_param_grads); there the outputs are held to 1e-4, the gradients to 2e-4 at the MEDIAN entry of every
    tensor and 5e-2 at the worst (a wrong formula moves every entry by O(1))."""
    from color_neus_amd import background as B
    lib = _lib(library)
    g = torch.Generator().manual_seed(2)
    R, MF = (8, 10) if small else (21, 19)
    o, d = _rays(R, g)
    zf = torch.sort(torch.rand(R, MF, generator=g) * 3.0 + 0.5, dim=-1).values
    zf[:, -4:] = zf[:, -4:] * 40.0                                     # far-away background samples (|p| >> 1) next to ones inside the unit ball
    kw = dict(D=4, W=64, multires=6, multires_view=2, skips=(1,)) if small else {}
    torch.manual_seed(11)
    ref = BO.NeRF(**kw).double()
    sdict = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    o64, d64, z64 = o.double().requires_grad_(True), d.double().requires_grad_(True), zf.double().requires_grad_(True)
    sd = 2.0 / 16
    a64, c64 = BO.render_outside(ref, o64, d64, z64, sd)
    ca, cc = torch.randn(R, MF, generator=g, dtype=torch.float64), torch.randn(R, MF, 3, generator=g, dtype=torch.float64)
    ((a64 * ca).sum() + (c64 * cc).sum()).backward()
    nerf = B.NeRF(**kw)
    nerf.load_state_dict({k: v.float() for k, v in sdict.items()})
    nerf = nerf.to(dev)
    on, dn, zn = o.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True), zf.to(dev).requires_grad_(True)
    alpha, color = B.Background.apply(lib, nerf.config(), sd, on, dn, zn, *nerf.ordered_params(lib))
    ((alpha * ca.float().to(dev)).sum() + (color * cc.float().to(dev)).sum()).backward()
    assert G.relerr(alpha.detach().cpu(), a64.detach()) < 1e-4 and G.relerr(color.detach().cpu(), c64.detach()) < 1e-4
    got = {"rays_o": on.grad, "rays_d": dn.grad, "z_feed": zn.grad, **{k: p.grad for k, p in nerf.named_parameters()}}
    want = {"rays_o": o64.grad, "rays_d": d64.grad, "z_feed": z64.grad, **{k: p.grad for k, p in ref.named_parameters()}}
    bad = {}
    for k, w in want.items():
        w = w.reshape(-1)
        e = (got[k].detach().cpu().double().reshape(-1) - w).abs() / max(float(w.abs().max()), 1e-300)
        if small:
            if not float(e.max()) < 2e-4:
                bad[k] = float(e.max())
        elif not (float(e.median()) < 2e-4 and float(e.max()) < 5e-2):
            bad[k] = (float(e.median()), float(e.max()))
    assert not bad, bad


def _composite(library, dev, color_type):
    from color_neus_amd import background as B
    lib = _lib(library)
    g = torch.Generator().manual_seed(3)
    R, M, n_out = 23, 20, 6
    MF = M + n_out
    o, d = _rays(R, g)
    z = torch.sort(torch.rand(R, M, generator=g) * 2.4 + 1.0, dim=-1).values          # straddles the unit sphere: inside and outside samples
    zf = torch.sort(torch.cat([z, torch.rand(R, n_out, generator=g) * 30 + 4.0], -1), -1).values
    sdf = torch.randn(R, M, generator=g) * 0.05
    grads = torch.nn.functional.normalize(torch.randn(R, M, 3, generator=g), dim=-1) * (1 + 0.2 * torch.randn(R, M, 1, generator=g))
    col, gcol = torch.rand(R, M, 3, generator=g), (torch.rand(R, M, 3, generator=g) if color_type else None)
    bga, bgc = torch.rand(R, MF, generator=g) * 0.3, torch.rand(R, MF, 3, generator=g)
    var = torch.tensor([0.35])
    bg_rgb = torch.tensor([0.2, 0.5, 0.7])
    sd, car = 2.0 / 16, 0.3
    names = ["rays_d", "z", "zf", "sdf", "grads", "col", "bga", "bgc", "var"] + (["gcol"] if color_type else [])
    vals = dict(rays_d=d, z=z, zf=zf, sdf=sdf, grads=grads, col=col, bga=bga, bgc=bgc, var=var, gcol=gcol)
    # oracle
    t64 = {k: vals[k].double().requires_grad_(True) for k in names}
    inv_s = torch.exp(t64["var"] * 10.0).clip(1e-6, 1e6)
    out64 = BO.composite_with_background("Color_NeuS" if color_type else "NeuS", o.double(), t64["rays_d"], t64["z"], sd, inv_s, t64["sdf"], t64["grads"], t64["col"],
                                         t64.get("gcol"), None, t64["bga"], t64["bgc"], t64["zf"], car, bg_rgb.double())
    keys = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "weights", "gradient_error", "depth"] + (["global_color"] if color_type else [])
    coefs = {k: torch.randn(out64[k].shape, generator=g, dtype=torch.float64) for k in keys}
    sum((out64[k] * coefs[k]).sum() for k in keys).backward()
    # native
    tn = {k: vals[k].to(dev).requires_grad_(True) for k in names}
    res = B.CompositeBg.apply(lib, sd, car, bg_rgb.to(dev), o.to(dev), tn["rays_d"], tn["z"], tn["zf"], tn["sdf"], tn["grads"], tn["col"], tn.get("gcol"), tn["bga"],
                              tn["bgc"], tn["var"])
    cn_ = [k for k in B._COMP_OUT if color_type or k != "global_color"] + ["inside_sphere", "eik_sums"]
    outn = dict(zip(cn_, res))
    sum((outn[k] * coefs[k].float().to(dev).reshape(outn[k].shape)).sum() for k in keys).backward()
    bad = {}
    for k in keys + ["inside_sphere"]:
        e = G.relerr(outn[k].detach().cpu().reshape(out64[k].shape), out64[k].detach())
        if not e < 1e-4:
            bad["out:" + k] = e
    for k in names:
        e = G.relerr(tn[k].grad.cpu().reshape(t64[k].grad.shape), t64[k].grad)
        if not e < 2e-4:
            bad["grad:" + k] = e
    assert not bad, bad


EMU = pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")


@EMU
@pytest.mark.parametrize("perturb", [False, True])
def test_outside_z_emu(perturb):
    _outside_z(N.EMU_LIB, "cpu", perturb)


@EMU
@pytest.mark.parametrize("small", [True, False])
def test_background_network_emu(small):
    _background(N.EMU_LIB, "cpu", small)


@EMU
@pytest.mark.parametrize("color_type", [False, True])
def test_composite_background_emu(color_type):
    _composite(N.EMU_LIB, "cpu", color_type)


@pytest.mark.gpu
@pytest.mark.parametrize("perturb", [False, True])
def test_outside_z_hip(perturb):
    _outside_z(None, "cuda:0", perturb)


@pytest.mark.gpu
@pytest.mark.parametrize("small", [True, False])
def test_background_network_hip(small):
    _background(None, "cuda:0", small)


@pytest.mark.gpu
@pytest.mark.parametrize("color_type", [False, True])
def test_composite_background_hip(color_type):
    _composite(None, "cuda:0", color_type)
