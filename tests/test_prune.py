"""Inference-only early-termination compaction (prune_eps > 0): colour / relight networks run only on samples with
compositing weight >= eps.  CPU: emulation build; GPU (-m gpu): HIP build (wavefront ballot + popcount prefix)."""
import os

import pytest
import torch

import _golden as G
import _native as N


def _check(library, device, name):
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    r = N.make_renderer(ocfg, P, library, device)
    o, d = torch.from_numpy(fx["rays_o"]).to(device), torch.from_numpy(fx["rays_d"]).to(device)
    near, far = torch.from_numpy(fx["jit:near"]).to(device), torch.from_numpy(fx["jit:far"]).to(device)
    z = torch.from_numpy(fx["jit:z_vals"]).to(device)
    eps = 1e-3
    with torch.no_grad():
        full = r(o, d, near, far, z_vals=z)
        pr = r(o, d, near, far, z_vals=z, prune_eps=eps)
    keep = full["weights"] >= eps
    assert 0 < int(keep.sum()) < keep.numel(), "test needs both kept and pruned samples"
    for k in ("weights", "depth", "weight_sum", "cdf_fine", "gradients", "gradient_error"):
        assert torch.equal(full[k], pr[k]), k                       # geometry side is untouched
    # every skipped sample changes the pixel by < eps
    n_pruned = (~keep).sum(-1, keepdim=True).float()
    assert bool(((full["color_fine"] - pr["color_fine"]).abs() <= n_pruned * eps + 1e-6).all())
    assert float((full["color_fine"] - pr["color_fine"]).abs().max()) < 2e-2
    if "delta_relight" in full:
        assert torch.allclose(pr["delta_relight"][keep], full["delta_relight"][keep], atol=1e-6)
        assert float(pr["delta_relight"][~keep].abs().max()) == 0.0
    # training through a pruned forward is refused
    o2 = o.clone().requires_grad_(True)
    out = r(o2, d, near, far, z_vals=z, prune_eps=eps)
    with pytest.raises(RuntimeError, match="inference-only"):
        out["color_fine"].sum().backward()


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp"])
def test_prune_emu(name):
    _check(N.EMU_LIB, "cpu", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp", "dtu_sharp"])
def test_prune_hip(name):
    _check(None, "cuda:0", name)
