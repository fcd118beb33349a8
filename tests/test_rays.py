"""Ray generation (the producer in front of the path) through the C ABI (cnr_gen_rays / cnr_gen_rays_backward)
against the golden vectors captured from the reference's ray_utils.py; CPU-emulation build here, HIP kernels under -m gpu."""
import os

import pytest
import torch

import _golden as G
import _native as N
import color_neus_amd as cn
from color_neus_amd import rays

emu = pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")


def _multicam(library, dev):
    fx = G.load("rays")
    c2w, focal, image, mask = (torch.from_numpy(fx[k]).to(dev) for k in ("c2w", "focal", "image", "mask"))
    t = lambda k: torch.from_numpy(fx[k])
    torch.manual_seed(5)
    o, d, rgb, ms = rays.get_rays_multicam(c2w, focal, image, 40, mask=mask, mask_rate=0.7, return_mask=True, normalize=True, library=library)
    assert torch.equal(rgb.cpu(), t("m:rgb")) and torch.equal(ms.cpu(), t("m:mask"))   # identical pixel choice
    assert torch.allclose(o.cpu(), t("m:o"), atol=1e-6) and torch.allclose(d.cpu(), t("m:d"), atol=1e-6)
    torch.manual_seed(5)
    o, d, rgb, ms = rays.get_rays_multicam(c2w, focal, image, 40, mask=None, normalize=False, opengl=True, library=library)
    assert ms is None and torch.equal(rgb.cpu(), t("nm:rgb"))
    assert torch.allclose(o.cpu(), t("nm:o"), atol=1e-6) and torch.allclose(d.cpu(), t("nm:d"), atol=1e-6)


def _at_and_nearfar(library, dev):
    fx = G.load("rays")
    c2w, focal = torch.from_numpy(fx["c2w"]).to(dev), torch.from_numpy(fx["focal"]).to(dev)
    t = lambda k: torch.from_numpy(fx[k])
    o, d = rays.get_rays_at(c2w[1], focal, 12, 17, normalize=True, library=library)
    assert o.shape == (12, 17, 3)
    assert torch.allclose(o.cpu(), t("at:o"), atol=1e-6) and torch.allclose(d.cpu(), t("at:d"), atol=1e-6)
    near, far = rays.near_far_from_sphere(t("m:o"), t("m:d"))
    assert torch.allclose(near, t("nf:near"), atol=1e-6) and torch.allclose(far, t("nf:far"), atol=1e-6)
    # the fused trainer front end: same pixels, normalised origins, near / far of the normalised rays
    image, mask = t("image").to(dev), t("mask").to(dev)
    origin, radius = torch.tensor([0.1, -0.2, 0.05]), 1.7
    torch.manual_seed(5)
    o2, d2, near2, far2, rgb2, ms2 = rays.rays_for_training(c2w, focal, image, 40, origin, radius, normalize=True, mask=mask, mask_rate=0.7,
                                                            return_mask=True, library=library)
    assert torch.equal(rgb2.cpu(), t("m:rgb")) and torch.equal(ms2.cpu(), t("m:mask"))
    want_o = (t("m:o") - origin) / radius
    assert torch.allclose(o2.cpu(), want_o, atol=1e-6) and torch.allclose(d2.cpu(), t("m:d"), atol=1e-6)
    wn, wf = rays.near_far_from_sphere(want_o, t("m:d"))
    assert torch.allclose(near2.cpu(), wn, atol=2e-6) and torch.allclose(far2.cpu(), wf, atol=2e-6)


def _torch_rays(c2w, focal, idx, H, W, normalize, opengl, origin, radius):
    """plain differentiable torch formula of the same rays (float64), the autograd reference of the backward kernel"""
    cam = torch.div(idx, H * W, rounding_mode="floor")
    pix = idx - cam * H * W
    py, px = torch.div(pix, W, rounding_mode="floor").double(), (pix % W).double()
    sgn = -1.0 if opengl else 1.0
    u = torch.stack([(px - W * 0.5) / focal[0], sgn * (py - H * 0.5) / focal[1], sgn * torch.ones_like(px)], -1)
    if normalize:
        u = u / u.norm(dim=-1, keepdim=True)
    d = (u[:, None, :] * c2w[cam, :3, :3]).sum(-1)
    o = (c2w[cam, :3, 3] - origin) / radius
    mid = -(o * d).sum(-1) / (d * d).sum(-1)
    return o, d, mid - 1.0, mid + 1.0


def _backward(library, dev):
    fx = G.load("rays")
    g = torch.Generator().manual_seed(2)
    H, W, n = 12, 17, 64
    for normalize, opengl in ((True, False), (False, True)):
        c2w = torch.from_numpy(fx["c2w"]).clone()
        focal = torch.from_numpy(fx["focal"]).clone()
        idx = torch.randint(0, 3 * H * W, (n,), generator=g)
        origin, radius = torch.tensor([0.1, -0.2, 0.05]), 1.7
        co = [torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)]
        c64, f64 = c2w.double().requires_grad_(True), focal.double().requires_grad_(True)
        outs = _torch_rays(c64, f64, idx, H, W, normalize, opengl, origin.double(), radius)
        sum((a * b.double()).sum() for a, b in zip(outs, co)).backward()
        cd, fd = c2w.to(dev).requires_grad_(True), focal.to(dev).requires_grad_(True)
        lib = rays._library(library)
        o, d, _, _, near, far = rays._generate(lib, idx.to(dev), n, cd, fd, H, W, normalize, opengl, origin=origin, radius=radius, want_nearfar=True)
        for a, b in zip((o, d, near, far), outs):
            assert torch.allclose(a.detach().cpu().double(), b.detach(), atol=2e-6)
        (sum((a * b.to(dev)).sum() for a, b in zip((o, d, near, far), co))).backward()
        assert float((cd.grad.cpu().double() - c64.grad).abs().max()) < 2e-5 * float(c64.grad.abs().max())
        assert float((fd.grad.cpu().double() - f64.grad).abs().max()) < 2e-5 * float(f64.grad.abs().max())
        assert float(cd.grad[:, 3].abs().max()) == 0.0     # bottom row of the pose matrices


def _bad_indices(library, dev):
    """A caller-supplied index outside [0, n_cams * H * W): the kernel reads nothing for it (the image / mask / pose reads would be out of
    bounds), that ray's outputs are NaN, its backward contributes nothing, and every other ray is untouched."""
    fx = G.load("rays")
    H, W = 12, 17
    c2w = torch.from_numpy(fx["c2w"]).to(dev).requires_grad_(True)
    focal = torch.from_numpy(fx["focal"]).to(dev)
    image, mask = torch.from_numpy(fx["image"]).to(dev), torch.from_numpy(fx["mask"]).to(dev)
    n_cams = c2w.shape[0]
    good = torch.tensor([0, 5, H * W + 3, n_cams * H * W - 1], dtype=torch.int64)
    bad = torch.tensor([0, -1, H * W + 3, n_cams * H * W, 5, 1 << 40], dtype=torch.int64)
    lib = rays._library(library)
    og, dg, rg, mg, ng, fg = rays._generate(lib, good.to(dev), 4, c2w, focal, H, W, True, False, image=image, mask=mask, want_nearfar=True)
    assert rays.bad_index_count(dev) == 0
    rays.raise_if_bad_indices(dev)
    ob, db, rb, mb, nb, fb = rays._generate(lib, bad.to(dev), 6, c2w, focal, H, W, True, False, image=image, mask=mask, want_nearfar=True)
    with pytest.raises(IndexError, match="3 pixel"):    # the device-side error counter: raised at the caller's next synchronisation point
        rays.raise_if_bad_indices(dev)
    assert rays.bad_index_count(dev) == 0                # (reset by the check)
    for t in (ob, db, rb, mb, nb, fb):
        t = t.detach().cpu().reshape(6, -1)
        assert bool(torch.isnan(t[[1, 3, 5]]).all()) and bool(torch.isfinite(t[[0, 2, 4]]).all())
    assert torch.equal(ob[[0, 2, 4]].detach().cpu(), og[[0, 2, 1]].detach().cpu()) and torch.equal(rb[[0, 2, 4]].cpu(), rg[[0, 2, 1]].cpu())
    # backward: cotangents of the bad rays are ignored (zeroed here so that NaN x 0 does not enter through the caller's own arithmetic)
    w = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0, 0.0], device=dev)[:, None]
    (torch.nan_to_num(ob) * w).sum().backward()
    g_bad = c2w.grad.clone()
    c2w.grad = None
    og[[0, 2, 1]].sum().backward()
    assert bool(torch.isfinite(g_bad).all()) and torch.allclose(g_bad, c2w.grad, atol=1e-6)


@emu
def test_out_of_range_pixel_indices_emu():
    _bad_indices(N.EMU_LIB, "cpu")


@pytest.mark.gpu
def test_out_of_range_pixel_indices_hip():
    _bad_indices(None, "cuda:0")


@emu
def test_get_rays_multicam_same_pixels_and_values_emu():
    _multicam(N.EMU_LIB, "cpu")


@emu
def test_get_rays_at_and_near_far_emu():
    _at_and_nearfar(N.EMU_LIB, "cpu")


@emu
def test_pose_and_focal_gradients_emu():
    _backward(N.EMU_LIB, "cpu")


@pytest.mark.gpu
def test_ray_generation_hip():
    _multicam(None, "cuda:0")
    _at_and_nearfar(None, "cuda:0")
    _backward(None, "cuda:0")


@emu
def test_argument_checks():
    lib = cn.load_library(N.EMU_LIB)
    with pytest.raises(AssertionError):
        rays.get_rays_multicam(torch.eye(4), torch.ones(2), torch.zeros(1, 4, 4, 3), 4, library=lib)      # single pose given to the multi-camera form
    with pytest.raises(AssertionError):
        rays.get_rays_at(torch.eye(4)[None], torch.ones(2), 4, 4, library=lib)
