"""CPU: pin the oracle (oracle/colorneus_oracle.py) to golden vectors captured from the reference.

Gates follow SURVEY.md 8(c):
  G1 sampler      : final z_vals vs golden, atol 1e-3 (the reference's own fp32-vs-fp64 band is 6.7e-4)
  G2 render_core  : all outputs + all parameter grads + d rays at the GOLDEN z_vals, rel 1e-4
  G3 end-to-end   : color_fine/depth/weight_sum/gradient_error/loss at rel 1e-4 (init regime)
"""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import colorneus_oracle as O

TOL = 1e-4  # relative (max-abs normalised), the north_star tolerance


def test_positional_encoding():
    fx = G.load("functions")
    for L in (4, 6):
        y = O.positional_encoding(torch.from_numpy(fx[f"pe{L}:x"]), L)
        assert torch.equal(y, torch.from_numpy(fx[f"pe{L}:y"]))


def test_inverse_sigmoid_edges():
    fx = G.load("functions")
    y = O.inverse_sigmoid(torch.from_numpy(fx["isig:x"]))
    assert torch.allclose(y, torch.from_numpy(fx["isig:y"]), rtol=1e-6, atol=0)


def test_sample_pdf_edge_cases():
    fx = G.load("functions")
    bins, w = torch.from_numpy(fx["spdf:bins"]), torch.from_numpy(fx["spdf:w"])
    for n in (16, 5):
        y = O.sample_pdf_det(bins, w, n)
        assert torch.allclose(y, torch.from_numpy(fx[f"spdf:out{n}"]), rtol=0, atol=1e-6)


def test_sample_pdf_random_draws():
    """det=False (ray_utils.py:135-136): the oracle at the reference's draws reproduces the reference's samples."""
    fx = G.load("sample_pdf_random")
    bins, w = torch.from_numpy(fx["bins"]), torch.from_numpy(fx["w"])
    for n in (16, 5):
        y = O.sample_pdf_det(bins, w, n, u=torch.from_numpy(fx[f"u{n}"]))
        assert torch.allclose(y, torch.from_numpy(fx[f"out{n}"]), rtol=0, atol=1e-6)


def test_up_sample_and_merge():
    fx = G.load("functions")
    cfg = O.tiny_config()
    P = G.prefixed(fx, "tinyw:")
    o, d, z, sdf = (torch.from_numpy(fx["ups:" + k]) for k in ("o", "d", "z", "sdf"))
    for i in range(4):
        nz = O.up_sample(o, d, z, sdf, 4, 64 * 2 ** i)
        assert torch.allclose(nz, torch.from_numpy(fx[f"ups:new_z_{i}"]), rtol=0, atol=1e-5)
    nz = O.up_sample(o, d, z, sdf, 4, 64.0)
    zc, sc = O.merge_z(P, cfg, o, d, z, nz, sdf, last=False)
    assert torch.allclose(zc, torch.from_numpy(fx["cat:z"]), atol=1e-5)
    assert torch.allclose(sc, torch.from_numpy(fx["cat:sdf"]), atol=1e-5)


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_networks(tag):
    fx = G.load("functions")
    cfg = O.tiny_config() if tag == "tiny" else G.mid_config()
    P = G.prefixed(fx, "tinyw:" if tag == "tiny" else "midw:")
    pts, dirs = torch.from_numpy(fx[f"{tag}:pts"]), torch.from_numpy(fx[f"{tag}:dirs"])
    sdf, feat, g = O.sdf_forward(P, cfg.sdf, pts, want_grad=True)
    ref = torch.from_numpy(fx[f"{tag}:sdf_out"])
    assert G.relerr(torch.cat([sdf, feat], -1), ref) < 1e-5
    assert G.relerr(g, fx[f"{tag}:sdf_grad"]) < 1e-5
    col = O.color_forward(P, cfg.color, pts, torch.from_numpy(fx[f"{tag}:sdf_grad"]), dirs, ref[:, 1:])
    assert G.relerr(col, fx[f"{tag}:color"]) < 1e-5
    if tag == "tiny":
        rel, drgb = O.relight_forward(P, cfg.relight, torch.from_numpy(fx["tiny:color"]), pts, dirs,
                                      torch.from_numpy(fx["tiny:sdf_grad"]))
        assert G.relerr(rel, fx["tiny:relit"]) < 1e-5 and G.relerr(drgb, fx["tiny:drgb"]) < 1e-5


def test_grid_and_vertex_colour():
    fx = G.load("functions")
    cfg = O.tiny_config()
    P = G.prefixed(fx, "tinyw:")
    lin = torch.linspace(-1.01, 1.01, 16)
    xx, yy, zz = torch.meshgrid(lin, lin, lin, indexing="ij")
    pts = torch.stack([xx, yy, zz], -1).reshape(-1, 3)
    u = -O.sdf_value(P, cfg.sdf, pts).reshape(16, 16, 16)
    assert G.relerr(u, fx["grid:u16"]) < 1e-5
    v = torch.from_numpy(fx["vcol:verts"]).float()
    sdf, feat, g = O.sdf_forward(P, cfg.sdf, v, want_grad=True)
    rgb = O.color_forward(P, cfg.color, v, g, -g, feat)
    assert G.relerr(rgb, fx["vcol:rgb"]) < 1e-5


def test_loss_counterpart():
    fx = G.prefixed(G.load("functions"), "loss:")
    out = dict(color_fine=fx["cf"], gradient_error=fx["ge"], weight_sum=fx["ws"], delta_relight=fx["dr"])
    l_on, _ = O.compute_loss(out, fx["gt"], fx["m"])
    l_off, _ = O.compute_loss(out, fx["gt"], None, lambda_mask=0.0, include_mask=False)
    assert abs(float(l_on) - float(fx["l_on"])) < 1e-6
    assert abs(float(l_off) - float(fx["l_off"])) < 1e-6


def _run_oracle(name, tag, fixed_z):
    fx = G.load(name)
    cfg, P = G.weights_of(name, fx)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    o = torch.from_numpy(fx["rays_o"]).requires_grad_(True)
    d = torch.from_numpy(fx["rays_d"]).requires_grad_(True)
    near, far = torch.from_numpy(fx[f"{tag}:near"]), torch.from_numpy(fx[f"{tag}:far"])
    t_rand = torch.from_numpy(fx[f"{tag}:t_rand"]) if f"{tag}:t_rand" in fx else None
    z = torch.from_numpy(fx[f"{tag}:z_vals"]) if fixed_z else None
    out = O.render(P, cfg, o, d, near, far, t_rand=t_rand, z_vals=z, **G.call_kwargs(fx))
    loss, _ = O.compute_loss(out, torch.from_numpy(fx["rgb_gt"]), torch.from_numpy(fx["mask"]))
    loss.backward()
    return fx, cfg, P, o, d, out, loss


# (*_anneal: cos_anneal_ratio 0.3 and a background colour, NeuS.py:294-302 -- the oracle branches of Color_NeuS.py:69-78, 104-106)
E2E = ["tiny_init", "tiny_sharp", "tiny_neus_sharp", "tiny_noimp_sharp", "dtu_init", "dtu_sharp", "neus_dtu_sharp", "tiny_sharp_anneal", "dtu_sharp_anneal"]
# round 6: the configuration branches no shipped YAML takes (tests/_golden.py VARIANTS), captured from the reference
E2E += list(G.VARIANTS)


@pytest.mark.parametrize("name", E2E)
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g1_sampler(name, tag):
    fx = G.load(name)
    cfg, P = G.weights_of(name, fx)
    o, d = torch.from_numpy(fx["rays_o"]), torch.from_numpy(fx["rays_d"])
    near, far = O.near_far_from_sphere(o, d)
    assert torch.allclose(near, torch.from_numpy(fx[f"{tag}:near"]), atol=1e-6)
    t_rand = torch.from_numpy(fx[f"{tag}:t_rand"]) if f"{tag}:t_rand" in fx else None
    z = O.sample_z(P, cfg, o, d, near, far, t_rand)
    assert G.check_g1(z, fx, tag) is None, G.check_g1(z, fx, tag)


@pytest.mark.parametrize("name", E2E)
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g2_render_core_at_golden_z(name, tag):
    fx, cfg, P, o, d, out, loss = _run_oracle(name, tag, fixed_z=True)
    for k in G.OUTPUT_KEYS:
        if f"{tag}:out_{k}" in fx:
            assert G.relerr(out[k].detach(), fx[f"{tag}:out_{k}"]) < TOL, k
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    bad = G.check_param_grads(fx, tag, {k: p.grad for k, p in P.items()}, TOL)
    # d variance is ONE number in which the per-ray terms cancel to a few percent of their size: a float32 evaluation with
    # another summation order than the reference's sits up to 2e-4 from float64 where the reference's own float32 run sits at 0.7e-4
    # (tiny_sharp_anneal / jit).  The float64 test below pins the restatement itself far tighter; here the scalar keeps the hard cap only
    bad = [b for b in bad if not (b[0] == "deviation_network.variance" and b[1] <= G.STRICT_TOL_CAP)]
    assert not bad, bad
    assert G.relerr(o.grad, fx[f"{tag}:grad_rays_o"]) < TOL
    assert G.relerr(d.grad, fx[f"{tag}:grad_rays_d"]) < TOL


@pytest.mark.parametrize("name", E2E)
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g2_oracle_in_float64_matches_the_reference_in_float64(name, tag):
    """The restatement itself, free of float32 round-off: the oracle evaluated in float64 at the fixture's sample positions against the
    REFERENCE code run in float64 at the same positions (tools/gen_golden.py stores its loss, every gradient tensor -- strided beyond 8192
    entries -- their sums, and d rays).  Two float64 evaluations of the same algorithm agree to ~1e-9; 1e-6 leaves room for the ReLU nets'
    summation order only.  This is what pins the oracle branches behind non-default call arguments (*_anneal fixtures) as well."""
    fx = G.load(name)
    cfg, P = G.weights_of(name, fx, dtype=torch.float64)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    t = lambda k: torch.from_numpy(fx[k]).double()
    o, d = t("rays_o").requires_grad_(True), t("rays_d").requires_grad_(True)
    noimp = cfg.n_importance == 0    # z is then a closed form of near / far and the float64 reference run used its own z (no override) ...
    if noimp:                        # ... from near / far evaluated in float64 (the fixture stores the float32 run's)
        near, far = (x.detach().requires_grad_(True) for x in O.near_far_from_sphere(o.detach(), d.detach()))
    else:
        near, far = t(f"{tag}:near").requires_grad_(True), t(f"{tag}:far").requires_grad_(True)
    t_rand = t(f"{tag}:t_rand") if (noimp and f"{tag}:t_rand" in fx) else None
    out = O.render(P, cfg, o, d, near, far, t_rand=t_rand, z_vals=None if noimp else t(f"{tag}:z_vals"), **G.call_kwargs(fx, dtype=torch.float64))
    loss, _ = O.compute_loss(out, t("rgb_gt"), t("mask"))
    loss.backward()
    # (without importance sampling the reference builds z from a float32 linspace even when it runs in float64, the oracle from a float64 one:
    # sample positions differ by 1e-8, gradients by 2e-6 of their scale)
    tol = 5e-5 if noimp else 1e-6   # (5e-5: d variance of tiny_noimp_sharp is a cancelled sum, 1.3e-5 there; every other tensor 2e-6)
    assert abs(float(loss.detach()) - float(fx[f"{tag}:f64:loss"])) < tol * abs(float(fx[f"{tag}:f64:loss"]))
    s = int(fx["grad_stride"])
    for k, p in P.items():
        ref = fx[f"{tag}:g64:{k}"]
        full = p.grad.reshape(-1)
        got = full[::(1 if full.numel() <= G.FULL_TENSOR_LIMIT else s)].numpy()
        den = max(float(fx[f"{tag}:gmax64:{k}"]), 1e-300)
        assert float(np.abs(got - ref).max()) / den < tol, (k, float(np.abs(got - ref).max()) / den)
        gabs = max(float(fx[f"{tag}:gabs64:{k}"]), 1e-300)
        assert abs(float(full.sum()) - float(fx[f"{tag}:gsum64:{k}"])) / gabs < tol and abs(float(full.abs().sum()) - gabs) / gabs < tol, k
    assert G.relerr(o.grad, fx[f"{tag}:f64:grad_rays_o"]) < tol and G.relerr(d.grad, fx[f"{tag}:f64:grad_rays_d"]) < tol
    if noimp:
        assert G.relerr(near.grad, fx[f"{tag}:f64:grad_near"]) < tol and G.relerr(far.grad, fx[f"{tag}:f64:grad_far"]) < tol


@pytest.mark.parametrize("name", ["tiny_init", "dtu_init"])
def test_g3_end_to_end_init_regime(name):
    fx, cfg, P, o, d, out, loss = _run_oracle(name, "jit", fixed_z=False)
    for k in ("color_fine", "depth", "weight_sum", "gradient_error"):
        assert G.relerr(out[k].detach(), fx[f"jit:out_{k}"]) < TOL, k
    assert abs(float(loss.detach()) - float(fx["jit:loss"])) < TOL * abs(float(fx["jit:loss"]))


def test_reference_ops_mode_matches_golden():
    """reference_ops=True (second SDF forward + autograd.grad(create_graph=True), the reference's executed work -- what the
    bench baselines time) gives the golden outputs and parameter gradients like the analytic sweep does."""
    fx = G.load("tiny_sharp")
    ocfg, P = G.weights_of("tiny_sharp", fx)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    t = lambda k: torch.from_numpy(fx[k])
    out = O.render(P, ocfg, t("rays_o"), t("rays_d"), t("jit:near"), t("jit:far"), z_vals=t("jit:z_vals"), reference_ops=True)
    for k in ("color_fine", "gradients", "weights", "delta_relight", "gradient_error"):
        assert G.relerr(out[k].detach().reshape(fx[f"jit:out_{k}"].shape), fx[f"jit:out_{k}"]) < 1e-4, k
    loss, _ = O.compute_loss(out, t("rgb_gt"), t("mask"))
    loss.backward()
    assert abs(float(loss.detach()) - float(fx["jit:loss"])) < 1e-6 * abs(float(fx["jit:loss"]))
    bad = G.check_param_grads(fx, "jit", {k: v.grad for k, v in P.items()})
    assert not bad, bad


@pytest.mark.parametrize("name", ["tiny_outside", "tiny_neus_outside"])
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_outside_oracle_against_reference_goldens(name, tag):
    """N_OUTSIDE = 8: oracle/background_oracle.py (background samples, NeRF network, render_core_outside, mixing in
    render_core) against the vectors captured from the imported reference (tools/gen_golden.py): outputs, loss, every parameter gradient
    incl. nerf.*, d rays.  This pins the restatement that the N_OUTSIDE tests of the native path use as their float64 reference."""
    from oracle import background_oracle as BO
    fx = G.load(name)
    cfg, P = G.weights_of(name, fx)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    t = lambda k: torch.from_numpy(fx[k])
    o, d = t("rays_o").requires_grad_(True), t("rays_d").requires_grad_(True)
    R = o.shape[0]
    t_out = None
    if tag == "jit":   # the fixture's run seeds the CPU generator with 2 and draws the foreground jitter first (tools/gen_golden.py)
        torch.manual_seed(2)
        torch.rand([R, 1])
        t_out = torch.rand([R, cfg.n_outside])
    out = BO.render(P, cfg, o, d, t(f"{tag}:near"), t(f"{tag}:far"), t(f"{tag}:z_vals"), t_out=t_out)
    for k in G.OUTPUT_KEYS:
        if f"{tag}:out_{k}" in fx:
            assert G.relerr(out[k].detach().reshape(fx[f"{tag}:out_{k}"].shape), fx[f"{tag}:out_{k}"]) < TOL, k
    loss, _ = O.compute_loss(out, t("rgb_gt"), t("mask"))
    loss.backward()
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    bad = G.check_param_grads(fx, tag, {k: p.grad for k, p in P.items()}, TOL)
    assert not bad, bad
    assert G.relerr(o.grad, fx[f"{tag}:grad_rays_o"]) < TOL
    assert G.relerr(d.grad, fx[f"{tag}:grad_rays_d"]) < TOL
