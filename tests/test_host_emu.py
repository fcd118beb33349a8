"""CPU: the complete host side (Python module -> ctypes -> C ABI -> orchestration -> operand views / epilogues /
per-point bodies) executed on the CPU-emulation build of the kernel layer and checked against the golden vectors.

This validates everything except the HIP kernels' own tiling / wavefront code, which the -m gpu tests cover."""
import os

import numpy as np
import pytest
import torch

import _golden as G
import _native as N

TOL = 1e-4

pytestmark = pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built (run __graft_entry__.build())")

E2E = ["tiny_init", "tiny_sharp", "tiny_neus_sharp", "tiny_noimp_sharp", "dtu_init", "dtu_sharp", "neus_dtu_sharp"]


@pytest.mark.parametrize("name", E2E)
def test_g2_render_core_forward_backward(name):
    tag = "jit"
    fx, r, out, loss, grads, o, d = N.run_native(name, tag, N.EMU_LIB, "cpu", fixed_z=True)
    for k in G.OUTPUT_KEYS:
        if f"{tag}:out_{k}" in fx:
            assert G.relerr(out[k].detach().reshape(fx[f"{tag}:out_{k}"].shape), fx[f"{tag}:out_{k}"]) < TOL, k
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    bad = G.check_param_grads(fx, tag, grads, TOL)
    assert not bad, bad
    assert G.relerr(o.grad, fx[f"{tag}:grad_rays_o"]) < TOL
    assert G.relerr(d.grad, fx[f"{tag}:grad_rays_d"]) < TOL


@pytest.mark.parametrize("name", ["tiny_init", "tiny_sharp", "tiny_neus_sharp"])
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g1_sampler(name, tag):
    fx, r, out, loss, grads, o, d = N.run_native(name, tag, N.EMU_LIB, "cpu", fixed_z=False, rays_grad=False)
    assert float((out["z_vals"] - torch.from_numpy(fx[f"{tag}:z_vals"])).abs().max()) < 1e-3


def test_g3_end_to_end_init():
    fx, r, out, loss, grads, o, d = N.run_native("tiny_init", "jit", N.EMU_LIB, "cpu", fixed_z=False, rays_grad=False)
    for k in ("color_fine", "depth", "weight_sum", "gradient_error"):
        assert G.relerr(out[k].detach().reshape(fx[f"jit:out_{k}"].shape), fx[f"jit:out_{k}"]) < TOL, k
    assert abs(float(loss.detach()) - float(fx["jit:loss"])) < TOL * abs(float(fx["jit:loss"]))
