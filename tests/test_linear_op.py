"""cnr_linear_forward / cnr_linear_backward (one plain nn.Linear (+ ReLU) on the render library's layer and weight-gradient kernels; the
shapes are those of the NeRF++ background network, fields.py:192-274, which cnr_background_forward chains internally) against torch.nn.functional.linear (+ ReLU) with autograd: every shape of the NeRF stack and ragged point counts."""
import os

import pytest
import torch

import _native as N

import ctypes as C


class HipLinear(torch.autograd.Function):
    """y = act(x W^T + b) through cnr_linear_forward / cnr_linear_backward (the render library's layer GEMM + weight-gradient GEMM)."""

    @staticmethod
    def forward(ctx, lib, x, weight, bias, relu):
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous().float()
        w, b = weight.detach().contiguous().float(), (bias.detach().contiguous().float() if bias is not None else None)
        n, k, n_out = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty(n, n_out, dtype=torch.float32, device=x2.device)
        if n > 0:
            nb = lib.lib.cnr_linear_scratch_bytes(n, k, n_out, 0)
            scratch = torch.empty(nb, dtype=torch.uint8, device=x2.device)
            stream = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream) if x2.is_cuda else C.c_void_p(0)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            lib.check(lib.lib.cnr_linear_forward(p(x2), n, k, p(w), p(b), n_out, int(relu), p(y), p(scratch), nb, stream), "cnr_linear_forward")
        ctx.lib, ctx.relu, ctx.has_bias, ctx.xshape = lib, bool(relu), bias is not None, x.shape
        ctx.save_for_backward(x2, w, y)
        return y.reshape(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        lib = ctx.lib
        n, k, n_out = x2.shape[0], x2.shape[1], w.shape[0]
        dy2 = dy.reshape(-1, n_out).contiguous().float()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        dW = torch.empty_like(w)
        db = torch.empty(n_out, dtype=torch.float32, device=w.device) if ctx.has_bias else None
        if n > 0:
            nb = lib.lib.cnr_linear_scratch_bytes(n, k, n_out, 1)
            scratch = torch.empty(nb, dtype=torch.uint8, device=x2.device)
            stream = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream) if x2.is_cuda else C.c_void_p(0)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            lib.check(lib.lib.cnr_linear_backward(p(x2), p(y), p(dy2), n, k, p(w), n_out, int(ctx.relu), p(dx), p(dW), p(db), p(scratch), nb, stream),
                      "cnr_linear_backward")
        else:
            dW.zero_()
            if db is not None:
                db.zero_()
        return None, (dx.reshape(ctx.xshape) if dx is not None else None), dW, db, None


def _embed(x, multires):
    """get_embedder(multires, input_dims=d): [x, sin(2^k x), cos(2^k x)]_k (PositionEncoding.py:51-76)."""
    out = [x]
    for k in range(multires):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]


# (k, n_out, relu): pts_linears[0], pts_linears[i], the skip layer, alpha / feature heads, the view branch, the rgb head (NeRF(), fields.py:228-248)
SHAPES = [(84, 256, True), (256, 256, True), (340, 256, True), (256, 1, False), (256, 256, False), (283, 128, True), (128, 3, False), (5, 7, True)]


def _check(library, device, sizes):
    import color_neus_amd as cn
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
