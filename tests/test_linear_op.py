"""cnr_linear_forward / cnr_linear_backward (the layers of the NeRF++ background network, fields.py:192-274, on the render library's layer and
weight-gradient kernels) against torch.nn.functional.linear (+ ReLU) with autograd: every shape of the NeRF stack and ragged point counts."""
import os

import pytest
import torch

import _native as N

# (k, n_out, relu): pts_linears[0], pts_linears[i], the skip layer, alpha / feature heads, the view branch, the rgb head (NeRF(), fields.py:228-248)
SHAPES = [(84, 256, True), (256, 256, True), (340, 256, True), (256, 1, False), (256, 256, False), (283, 128, True), (128, 3, False), (5, 7, True)]


def _check(library, device, sizes):
    import color_neus_amd as cn
    from color_neus_amd.background import HipLinear
    lib = cn.load_library(library)
    g = torch.Generator().manual_seed(0)
    for n in sizes:
        for k, n_out, relu in SHAPES:
            x = torch.randn(n, k, generator=g).to(device).requires_grad_(True)
            w = (torch.randn(n_out, k, generator=g) / k ** 0.5).to(device).requires_grad_(True)
            b = (torch.randn(n_out, generator=g) * 0.1).to(device).requires_grad_(True)
            dy = torch.randn(n, n_out, generator=g).to(device)
            ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
            if relu:
                ref = torch.relu(ref)
            gx, gw, gb = torch.autograd.grad(ref, [x, w, b], dy.double())
            y = HipLinear.apply(lib, x, w, b, relu)
            hx, hw, hb = torch.autograd.grad(y, [x, w, b], dy)
            for name, a, r in (("y", y, ref), ("dx", hx, gx), ("dW", hw, gw), ("db", hb, gb)):
                den = max(float(r.abs().max()), 1e-30)
                err = float((a.double() - r).abs().max()) / den
                assert err < 2e-5, (n, k, n_out, relu, name, err)


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_linear_op_emu():
    _check(N.EMU_LIB, "cpu", [1, 33, 200])


@pytest.mark.gpu
def test_linear_op_hip():
    _check(None, "cuda:0", [1, 33, 1280, 5000])
