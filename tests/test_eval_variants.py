"""The evaluation entry points (cnr_sdf_grid, cnr_sdf_eval, cnr_vertex_color: NeuS.extract_fields / sdf / extract_color, NeuS.py:14-64) on the
network-configuration branches of tests/_golden.py VARIANTS -- WEIGHT_NORM False, MODE no_normal (view_dirs = -gradients into the colour
network), SQUEEZE_OUT False, the skip connection at another layer, TWO skip connections, MULTIRES 4 / SCALE 2 -- against the oracle, which
tests/test_oracle_golden.py pins to the reference on the same configurations.  (The render path of these configurations is held to the
reference goldens by the G1 / G2 gates; this file covers the other entry points that take a cnr_config.)  CPU: emulation build; GPU: HIP build."""
import os

import numpy as np
import pytest
import torch

import _golden as G
import _native as N
from oracle import colorneus_oracle as O

NAMES = ["tiny_nown_skip2", "tiny_twoskip", "tiny_nosq", "tiny_neus_nonormal", "dtu_twoskip", "dtu_nown_skip6", "neus_dtu_nonormal"]


def _check(name, library, device):
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    r = N.make_renderer(ocfg, P, library, device)
    g = torch.Generator().manual_seed(17)
    pts = torch.randn(300, 3, generator=g)
    pts = pts / pts.norm(dim=-1, keepdim=True) * (0.5 + 0.1 * torch.randn(300, 1, generator=g))   # around the r = 0.5 surface of the trained-like weights
    sdf, feat, grad = O.sdf_forward(P, ocfg.sdf, pts, want_grad=True)
    # sdf() on its own and the lattice
    got = r.sdf(pts.to(device)).cpu()
    assert G.relerr(got[:, 0], sdf.detach().reshape(-1)) < 1e-4, name
    res = 12
    u = r.extract_fields([-1.01] * 3, [1.01] * 3, device, res).cpu()
    lin = torch.linspace(-1.01, 1.01, res)
    grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), dim=-1).reshape(-1, 3)
    want_u = -O.sdf_forward(P, ocfg.sdf, grid)[0].detach().reshape(res, res, res)
    assert G.relerr(u, want_u) < 1e-4, name
    # vertex colours: color_network(pts, g, -g, feat) (NeuS.py:44-64); squeeze_out False leaves them unbounded
    want = O.color_forward(P, ocfg.color, pts, grad, -grad, feat).detach()
    rgb = r.extract_color(pts.numpy(), device)
    assert G.relerr(torch.from_numpy(np.asarray(rgb)), want) < 1e-4, name


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
@pytest.mark.parametrize("name", NAMES)
def test_eval_entry_points_on_config_variants_emu(name):
    _check(name, N.EMU_LIB, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_eval_entry_points_on_config_variants_hip(name):
    _check(name, None, "cuda:0")
