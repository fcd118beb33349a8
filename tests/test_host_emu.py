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

E2E = ["tiny_init", "tiny_sharp", "tiny_neus_sharp", "tiny_noimp_sharp", "dtu_init", "dtu_sharp", "dtu_noimp_sharp", "neus_dtu_sharp", "tiny_sharp_anneal", "dtu_sharp_anneal"]
# round 6: the configuration branches no shipped YAML takes (tests/_golden.py VARIANTS), captured from the reference
E2E += list(G.VARIANTS)


@pytest.mark.parametrize("name", E2E)
def test_g2_render_core_forward_backward(name):
    tag = "jit"
    noimp = "noimp" in name   # no importance sampling: z is a closed form of near / far, which then carry gradients (NeuS.py:311-313)
    res = N.run_native(name, tag, N.EMU_LIB, "cpu", fixed_z=not noimp, nearfar_grad=noimp)
    fx, r, out, loss, grads, o, d = res[:7]
    bad = G.check_outputs(fx, tag, out, TOL)
    assert not bad, bad
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    bad = G.check_param_grads(fx, tag, grads)
    assert not bad, bad
    checks = [("grad_rays_o", o.grad), ("grad_rays_d", d.grad)] + ([("grad_near", res[7].grad), ("grad_far", res[8].grad)] if noimp else [])
    for key, got in checks:
        assert G.check_input_grad(fx, tag, key, got) is None, G.check_input_grad(fx, tag, key, got)


def test_variance_gradient_population():
    """deviation_network.variance is ONE number per run: the 1.5x-of-the-reference's-float32-error rule is held on the population of all
    fixture runs (tests/_golden.py: scalar_tolerance / check_scalar_population)."""
    pairs = []
    for name in E2E:
        for tag in ("det", "jit"):
            if "noimp" in name:
                continue
            res = N.run_native(name, tag, N.EMU_LIB, "cpu", fixed_z=True)
            pairs.append(G.scalar_error(res[0], tag, res[4]))
    assert G.check_scalar_population(pairs) is None, G.check_scalar_population(pairs)


@pytest.mark.parametrize("name", ["tiny_init", "tiny_sharp", "tiny_neus_sharp"])
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g1_sampler(name, tag):
    fx, r, out, loss, grads, o, d = N.run_native(name, tag, N.EMU_LIB, "cpu", fixed_z=False, rays_grad=False)
    assert G.check_g1(out["z_vals"], fx, tag) is None, G.check_g1(out["z_vals"], fx, tag)


def test_g3_end_to_end_init():
    fx, r, out, loss, grads, o, d = N.run_native("tiny_init", "jit", N.EMU_LIB, "cpu", fixed_z=False, rays_grad=False)
    for k in ("color_fine", "depth", "weight_sum", "gradient_error"):
        assert G.relerr(out[k].detach().reshape(fx[f"jit:out_{k}"].shape), fx[f"jit:out_{k}"]) < TOL, k
    assert abs(float(loss.detach()) - float(fx["jit:loss"])) < TOL * abs(float(fx["jit:loss"]))


@pytest.mark.parametrize("name", ["tiny_outside", "tiny_neus_outside"])
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_nerfpp_background(name, tag):
    """N_OUTSIDE = 8 through the product route: background samples, the NeRF++ network (encodings, skip concat, heads),
    density -> alpha, inside / outside mixing and the compositing over M + N_OUTSIDE samples all behind the C ABI (cnr_outside_z,
    cnr_background_*, cnr_composite_background_*; emulation build here); outputs, loss and every parameter gradient -- nerf.* included --
    against the reference."""
    fx, r, out, loss, grads, o, d = N.run_native(name, tag, N.EMU_LIB, "cpu", fixed_z=True)
    assert out["weights"].shape[1] == r.rcfg.n_total + 8
    for k in G.OUTPUT_KEYS:
        if f"{tag}:out_{k}" in fx:
            assert G.relerr(out[k].detach().reshape(fx[f"{tag}:out_{k}"].shape), fx[f"{tag}:out_{k}"]) < TOL, k
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    assert any(k.startswith("nerf.") for k in grads)
    # the ray-sharded objective needs the two eikonal sums of the shard (parallel.sharded_loss): same ratio as gradient_error
    es = out["eik_sums"].detach()
    assert es.shape == (2,) and abs(float(es[0] / (es[1] + 1e-5)) - float(out["gradient_error"].detach())) < 1e-6
    bad = G.check_param_grads(fx, tag, grads)
    assert not bad, bad
    for key, got in (("grad_rays_o", o.grad), ("grad_rays_d", d.grad)):
        assert G.check_input_grad(fx, tag, key, got) is None, G.check_input_grad(fx, tag, key, got)


def test_c5_dtu_size_lattice_and_vertex_colours_emu():
    """BASELINE config C5 at the DTU network size through the host path (chunking, lattice indexing, colour-chain plumbing)."""
    from oracle import colorneus_oracle as O
    fx = G.load("functions")
    ocfg = O.dtu_config()
    P = O.init_params(ocfg, seed=0, dtype=torch.float32, trained_like=True)
    r = N.make_renderer(ocfg, P, N.EMU_LIB, "cpu")
    u = r.extract_fields([-1.01] * 3, [1.01] * 3, "cpu", 32)
    assert G.relerr(u, fx["c5:u32"]) < TOL
    rgb = r.extract_color(fx["c5:verts"], "cpu")
    assert float(np.abs(rgb - fx["c5:rgb"]).max()) < TOL
