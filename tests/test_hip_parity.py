"""GPU (MI355X): parity of the HIP path against the golden vectors captured from the reference and against the oracle.

Everything goes through the product route: nn.Module -> autograd.Function -> ctypes -> C ABI -> HIP kernels.
Gates: G1 sampler z (atol 1e-3), G2 render_core at golden z (rel 1e-4: 12 outputs, all parameter grads,
d rays), G3 end-to-end in the init regime (rel 1e-4)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _golden as G
import _native as N

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"

E2E = ["tiny_init", "tiny_sharp", "tiny_neus_sharp", "tiny_noimp_sharp", "dtu_init", "dtu_sharp", "dtu_noimp_sharp", "neus_dtu_sharp",
       "tiny_sharp_anneal", "dtu_sharp_anneal"]   # (*_anneal: cos_anneal_ratio 0.3 + background_rgb, reference goldens of NeuS.py:294-302)
# round 6: the configuration branches no shipped YAML takes (WEIGHT_NORM / SQUEEZE_OUT / INCLUDE_GRAD / INV_SIGMOID False, MODE no_normal,
# Y_IN_LAYER 2 and == N_LAYERS, the skip connection at another layer) at the tiny size and at the DTU widths -- reference goldens, tests/_golden.py VARIANTS
E2E += list(G.VARIANTS)


def test_library_is_hip():
    import color_neus_amd as cn
    assert cn.load_library().backend == "hip-gfx950"


def _g2(name, tag):
    noimp = "noimp" in name   # no importance sampling (BASELINE C2): z is a closed form of near / far, which then carry gradients
    res = N.run_native(name, tag, None, DEV, fixed_z=not noimp, nearfar_grad=noimp)
    fx, r, out, loss, grads, o, d = res[:7]
    bad = G.check_outputs(fx, tag, out, TOL)
    assert not bad, bad
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    checks = [("grad_rays_o", o.grad), ("grad_rays_d", d.grad)] + ([("grad_near", res[7].grad), ("grad_far", res[8].grad)] if noimp else [])
    return fx, grads, checks


@pytest.mark.parametrize("name", E2E)
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g2_render_core_forward_backward(name, tag):
    """12 outputs and the loss to 1e-4; every parameter-gradient tensor and d rays (d near / d far without importance sampling) at its
    OWN scale against the reference's float64 run (tests/_golden.py: check_param_grads)."""
    fx, grads, checks = _g2(name, tag)
    bad = G.check_param_grads(fx, tag, grads, strict=True)
    assert not bad, bad
    for key, got in checks:
        assert G.check_input_grad(fx, tag, key, got, strict=True) is None, G.check_input_grad(fx, tag, key, got, strict=True)


def test_variance_gradient_population():
    """deviation_network.variance is ONE number per run -- a cancelling sum that carries the float32 round-off of the network outputs it is
    made of.  The 1.5x-of-the-reference's-own-float32-error rule is held on the population of all fixture runs (RMS over fixtures x tags:
    tests/_golden.py scalar_tolerance / check_scalar_population); each single run keeps max(3e-4, 3x) under the cap inside the G2 gate."""
    pairs = []
    for name in E2E:
        for tag in ("det", "jit"):
            if "noimp" in name:
                continue
            res = N.run_native(name, tag, None, DEV, fixed_z=True)
            pairs.append(G.scalar_error(res[0], tag, res[4]))
    assert G.check_scalar_population(pairs) is None, G.check_scalar_population(pairs)


@pytest.mark.parametrize("name", ["dtu_sharp", "dtu_init", "neus_dtu_sharp"])
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g2_every_gradient_entry_against_live_float64_oracle(name, tag):
    """The fixtures hold every 97th entry of the large gradient tensors; here EVERY entry of every parameter gradient of the DTU-size
    fixtures is compared (strict gate) with the oracle evaluated in float64 on the same inputs at the golden z.  The oracle itself is
    pinned to the reference's float64 run on the stored entries by tests/test_oracle_golden.py."""
    from oracle import colorneus_oracle as O
    fx, grads, _ = _g2(name, tag)
    t = lambda k: torch.from_numpy(fx[k])
    ref = {}
    for dt in (torch.float64, torch.float32):
        cfg, P = G.weights_of(name, fx, dtype=dt)
        P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        out = O.render(P, cfg, t("rays_o").to(dt), t("rays_d").to(dt), t(f"{tag}:near").to(dt), t(f"{tag}:far").to(dt), z_vals=t(f"{tag}:z_vals").to(dt))
        loss, _ = O.compute_loss(out, t("rgb_gt").to(dt), t("mask").to(dt))
        loss.backward()
        ref[dt] = {k: v.grad for k, v in P.items()}
    # the live float64 run must itself sit on the stored float64 reference entries (guards this test's own set-up)
    assert not G.check_param_grads(fx, tag, ref[torch.float64], strict=True)
    got = {(k[len("renderer."):] if k.startswith("renderer.") else k): v for k, v in grads.items()}
    bad = G.check_grads_full(ref[torch.float64], ref[torch.float32], got, strict=True)
    assert not bad, bad


@pytest.mark.parametrize("name", ["tiny_outside", "tiny_neus_outside"])
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_nerfpp_background(name, tag):
    """N_OUTSIDE = 8 on the HIP library end to end: background samples, the NeRF++ network (encodings, skip concat, heads),
    density -> alpha, inside / outside mixing and the compositing over M + N_OUTSIDE samples behind the C ABI (cnr_outside_z, cnr_background_*,
    cnr_composite_background_*); strict gate on every parameter gradient, nerf.* included."""
    fx, r, out, loss, grads, o, d = N.run_native(name, tag, None, DEV, fixed_z=True)
    for k in G.OUTPUT_KEYS:
        if f"{tag}:out_{k}" in fx:
            ref = fx[f"{tag}:out_{k}"]
            assert G.relerr(out[k].detach().cpu().reshape(ref.shape), ref) < TOL, k
    assert abs(float(loss.detach()) - float(fx[f"{tag}:loss"])) < TOL * abs(float(fx[f"{tag}:loss"]))
    bad = G.check_param_grads(fx, tag, grads, strict=True)
    assert not bad, bad
    for key, got in (("grad_rays_o", o.grad), ("grad_rays_d", d.grad)):
        assert G.check_input_grad(fx, tag, key, got, strict=True) is None, G.check_input_grad(fx, tag, key, got, strict=True)


def test_param_grad_error_table():
    """The per-tensor error table of the HIP path on the DTU-size fixtures (what the gate above condenses); written to
    gpurun_out/ so that it can be committed under profiles/."""
    lines = []
    for name in ("dtu_sharp", "dtu_init", "dtu_noimp_sharp", "neus_dtu_sharp"):
        for tag in ("det", "jit"):
            fx, grads, checks = _g2(name, tag)
            rows = G.param_grad_table(fx, tag, grads, strict=True)
            lines.append(G.format_grad_table(f"{name} / {tag}: HIP parameter gradients vs the reference's float64 run (own scale per tensor)", rows))
            for key, got in checks:
                ref64 = fx[f"{tag}:f64:{key}"]
                e = float(np.abs(got.detach().cpu().double().numpy().reshape(ref64.shape) - ref64).max()) / float(np.abs(ref64).max())
                lines.append("%-42s %8d %10.2e" % (key, ref64.size, e))
            lines.append("")
            assert max(r[2] for r in rows) <= G.STRICT_TOL_CAP
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "param_grad_error_table.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", E2E)
@pytest.mark.parametrize("tag", ["det", "jit"])
def test_g1_sampler(name, tag):
    fx, r, out, loss, grads, o, d = N.run_native(name, tag, None, DEV, fixed_z=False, rays_grad=False)
    z = out["z_vals"].cpu()
    assert G.check_g1(z, fx, tag) is None, G.check_g1(z, fx, tag)
    assert bool((z[:, 1:] >= z[:, :-1]).all()), "z_vals must be sorted"


@pytest.mark.parametrize("name", ["tiny_init", "dtu_init"])
def test_g3_end_to_end_init(name):
    fx, r, out, loss, grads, o, d = N.run_native(name, "jit", None, DEV, fixed_z=False, rays_grad=False)
    for k in ("color_fine", "depth", "weight_sum", "gradient_error"):
        ref = fx[f"jit:out_{k}"]
        assert G.relerr(out[k].detach().cpu().reshape(ref.shape), ref) < TOL, k
    assert abs(float(loss.detach()) - float(fx["jit:loss"])) < TOL * abs(float(fx["jit:loss"]))


def test_deterministic_two_runs():
    a = N.run_native("dtu_sharp", "jit", None, DEV, fixed_z=True)
    b = N.run_native("dtu_sharp", "jit", None, DEV, fixed_z=True)
    assert torch.equal(a[2]["color_fine"], b[2]["color_fine"])
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k


def test_against_oracle_larger_batch():
    """HIP vs the oracle evaluated in float64 on the host (the rounding-free value of the same algorithm) on 96 rays x 128
    samples, DTU-size network, trained-like weights (inv_s = 665: per-sample weights are the most rounding-sensitive
    outputs; two fp32 implementations differ from each other by more than either differs from float64)."""
    from oracle import colorneus_oracle as O
    ocfg = O.dtu_config()
    P = O.init_params(ocfg, seed=5, trained_like=True)
    P64 = {k: v.double() for k, v in P.items()}
    g = torch.Generator().manual_seed(9)
    R = 96
    o = torch.randn(R, 3, generator=g); o = o / o.norm(dim=-1, keepdim=True) * 2.7
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.3 - o, dim=-1)
    near, far = O.near_far_from_sphere(o, d)
    t_rand = torch.rand(R, 1, generator=g)
    z32 = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    r = N.make_renderer(ocfg, P, None, DEV)
    orig = torch.rand
    try:
        torch.rand = lambda *a, **k: t_rand.clone()
        out = r(o.to(DEV), d.to(DEV), near.to(DEV), far.to(DEV))
    finally:
        torch.rand = orig
    assert float((out["z_vals"].cpu() - z32).abs().max()) < 1e-3                      # G1
    oo = O.render(P64, ocfg, o.double(), d.double(), near.double(), far.double(), z_vals=z32.double())
    out2 = r(o.to(DEV), d.to(DEV), near.to(DEV), far.to(DEV), z_vals=z32.to(DEV))      # G2 at identical z
    # per-sample weights/cdf at inv_s = 665 are not reproducible to 1e-4 in fp32 by ANY implementation: the plain-torch fp32
    # evaluation of the same algorithm (= the reference's arithmetic) is 1.3e-4 away from float64 on this very input.
    # Their tolerance is therefore calibrated per tensor: 3 x the float32 oracle's own distance from float64 on this input, at least 1e-4, at most 5e-4.
    o32 = O.render(P, ocfg, o, d, near, far, z_vals=z32)
    cal = {k: min(5e-4, max(TOL, 3.0 * G.relerr(o32[k].detach().double(), oo[k].detach()))) for k in ("weights", "weight_max", "cdf_fine")}
    errs = {k: G.relerr(out2[k].detach().cpu().reshape(oo[k].shape), oo[k].detach()) for k in G.OUTPUT_KEYS}
    bad = {k: (e, cal.get(k, TOL)) for k, e in errs.items() if not e < cal.get(k, TOL)}
    assert not bad, bad


def test_results_do_not_depend_on_scratch_contents():
    """Context / backward-scratch buffers come from torch.empty: fill them with NaN bit patterns and require bit-identical
    outputs and gradients (pad columns that GEMMs read must be written by the library itself, never assumed zero)."""
    ref = N.run_native("dtu_sharp", "jit", None, DEV, fixed_z=True)
    orig_empty = torch.empty

    def poisoned_empty(*a, **k):
        t = orig_empty(*a, **k)
        if t.dtype == torch.uint8 and t.numel() > 4096:
            t.fill_(0xFF)
        return t
    try:
        torch.empty = poisoned_empty
        got = N.run_native("dtu_sharp", "jit", None, DEV, fixed_z=True)
    finally:
        torch.empty = orig_empty
    for k in G.OUTPUT_KEYS:
        if k in ref[2] and ref[2][k] is not None:
            assert torch.equal(ref[2][k], got[2][k]), k
    for k in ref[4]:
        assert torch.equal(ref[4][k], got[4][k]), k


@pytest.mark.parametrize("switch", ["CNR_DISABLE_WS", "CNR_WS_GENERIC", "CNR_DW_BF16", "CNR_DW_FP32", "CNR_WS_SERP=0", "CNR_WS_NOSTREAM",
                                    "CNR_NO_FUSED", "CNR_NO_CHAIN_FWD", "CNR_NO_CHAIN_SDF", "CNR_CHAIN_GRAD", "CNR_NO_SWEEP0", "CNR_NO_NARROW_BWD", "CNR_NO_NARROW_DX"])
def test_fallback_kernels_keep_parity(switch):
    """The debugging switches select the fallback kernels (FP32-MFMA layer GEMM, interpreted weight-stationary kernel, split-bf16 and
    FP32-MFMA weight-gradient tiles, one walk direction for every layer launch, the general layer kernel instead of its stream form, the
    per-layer launches instead of the chain-fused forward kernels, separate layer + weight-gradient launches for the narrow-input layers) or, for
    CNR_CHAIN_GRAD, the opt-in chain-fused gradient chain.  They are
    read once per process, so the G2 gate runs in a child process."""
    name, _, val = switch.partition("=")
    env = dict(os.environ, **{name: val or "1"})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_g2_render_core_forward_backward and sharp and det"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _oracle_batch(R, seed):
    from oracle import colorneus_oracle as O
    ocfg = O.dtu_config()
    P = O.init_params(ocfg, seed=5, trained_like=True)
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(R, 3, generator=g); o = o / o.norm(dim=-1, keepdim=True) * 2.7
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.3 - o, dim=-1)
    near, far = O.near_far_from_sphere(o, d)
    t_rand = torch.rand(R, 1, generator=g)
    gt = torch.rand(R, 3, generator=g)
    mask = (torch.rand(R, generator=g) > 0.3).float()
    return O, ocfg, P, o, d, near, far, t_rand, gt, mask


def test_parameter_gradients_against_float64_oracle_larger_batch():
    """All parameter gradients of one training loss on 160 rays x 128 samples (DTU-size network, inv_s = 665) against the
    oracle evaluated in float64 at the same z: covers the split-f16 weight-gradient tiles, their per-point scales and the
    skinny strips on an input that is not one of the golden fixtures."""
    import color_neus_amd as cn
    O, ocfg, P, o, d, near, far, t_rand, gt, mask = _oracle_batch(160, seed=21)
    z32 = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    oo = O.render(P64, ocfg, o.double(), d.double(), near.double(), far.double(), z_vals=z32.double())
    l64, _ = O.compute_loss(oo, gt.double(), mask.double())
    l64.backward()
    r = N.make_renderer(ocfg, P, None, DEV)
    out = r(o.to(DEV), d.to(DEV), near.to(DEV), far.to(DEV), z_vals=z32.to(DEV))
    loss, _ = cn.compute_loss(out, gt.to(DEV), mask.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(l64.detach())) < 2e-4 * abs(float(l64.detach()))
    got = dict(r.named_parameters())
    ref = {k: v.grad for k, v in P64.items()}
    # the same algorithm in plain float32 (= the reference's arithmetic): its distance from float64 calibrates the tolerance per tensor
    P32 = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    o32 = O.render(P32, ocfg, o, d, near, far, z_vals=z32)
    l32, _ = O.compute_loss(o32, gt, mask)
    l32.backward()
    bad = []
    for k, gr in ref.items():
        name = k if k in got else "renderer." + k
        den = float(gr.abs().max())
        e = ((got[name].grad.detach().cpu().double() - gr).abs() / den).reshape(-1)
        spread = float((P32[k].grad.double() - gr).abs().max()) / den
        lim = min(G.STRICT_TOL_CAP, G.grad_tolerance(spread, True))
        allowed = G._allowed(e.numel(), True)
        bulk = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        if not (float(e.max()) <= G.STRICT_TOL_CAP and bulk <= lim):   # own scale: hard cap on every entry + bulk tolerance (tests/_golden.py, strict gate)
            bad.append((k, float(e.max()), bulk, lim))
    assert not bad, bad


def test_non_finite_input_poisons_the_weight_gradients():
    """A NaN ray must not be dropped silently by the per-point scaling of the weight-gradient GEMM."""
    import color_neus_amd as cn
    O, ocfg, P, o, d, near, far, t_rand, gt, mask = _oracle_batch(64, seed=3)
    o[5, 1] = float("nan")
    r = N.make_renderer(ocfg, P, None, DEV)
    z = torch.linspace(0.5, 4.0, ocfg.n_samples + ocfg.n_importance).repeat(64, 1)
    out = r(o.to(DEV), d.to(DEV), near.to(DEV), far.to(DEV), z_vals=z.to(DEV))
    loss, _ = cn.compute_loss(out, gt.to(DEV), mask.to(DEV))
    loss.backward()
    assert not bool(torch.isfinite(loss.detach()))
    # (the ReLU stacks clamp NaN to 0 like fmaxf does, so only the SDF network -- softplus -- is required to carry the poison)
    finite = [n for n, p in r.named_parameters() if "sdf_network" in n and p.grad is not None and p.grad.numel() > 1024
              and bool(torch.isfinite(p.grad).all())]
    assert not finite, finite


def test_sdf_grid_and_vertex_colour():
    fx = G.load("functions")
    from oracle import colorneus_oracle as O
    ocfg = O.tiny_config()
    P = G.prefixed(fx, "tinyw:")
    r = N.make_renderer(ocfg, P, None, DEV)
    u = r.extract_fields([-1.01] * 3, [1.01] * 3, DEV, 16).cpu()
    assert G.relerr(u, fx["grid:u16"]) < TOL
    rgb = r.extract_color(fx["vcol:verts"], DEV)
    assert G.relerr(rgb, fx["vcol:rgb"]) < TOL
    pts = torch.from_numpy(fx["tiny:pts"]).to(DEV)
    s = r.sdf(pts).cpu()
    assert G.relerr(s[:, 0], fx["tiny:sdf_out"][:, 0]) < TOL


def _c5_renderer(device, library=None):
    fx = G.load("functions")
    from oracle import colorneus_oracle as O
    ocfg = O.dtu_config()
    P = O.init_params(ocfg, seed=0, dtype=torch.float32, trained_like=True)
    assert abs(O.params_checksum(P) - float(fx["c5:weight_checksum"])) <= 1e-9 * abs(float(fx["c5:weight_checksum"]))
    return fx, N.make_renderer(ocfg, P, library, device)


def test_c5_dtu_size_lattice_and_vertex_colours():
    """BASELINE config C5 on the DTU-size network (weight-stationary kernels): a 32^3 lattice of extract_fields and 1000 vertex colours
    of extract_color against the vectors captured from the reference (NeuS.py:14-64)."""
    fx, r = _c5_renderer(DEV)
    u = r.extract_fields([-1.01] * 3, [1.01] * 3, DEV, 32).cpu()
    assert u.shape == (32, 32, 32)
    assert G.relerr(u, fx["c5:u32"]) < TOL
    rgb = r.extract_color(fx["c5:verts"], DEV)
    assert rgb.shape == (1000, 3)
    assert float(np.abs(rgb - fx["c5:rgb"]).max()) < TOL      # colours are in [0, 1]


def test_eval_paths_do_not_depend_on_scratch_contents():
    """extract_fields / extract_color / sdf with NaN-filled scratch buffers are bit-identical to a clean run (the GEMMs read padded
    columns: every one of them must be written by the library itself)."""
    fx, r = _c5_renderer(DEV)
    pts = torch.from_numpy(fx["c5:verts"]).to(DEV)

    def run():
        return (r.extract_fields([-1.01] * 3, [1.01] * 3, DEV, 24).clone(), torch.from_numpy(r.extract_color(fx["c5:verts"], DEV)),
                r.sdf(pts).clone())
    ref = run()
    orig_empty = torch.empty

    def poisoned_empty(*a, **k):
        t = orig_empty(*a, **k)
        if t.dtype == torch.uint8 and t.numel() > 4096:
            t.fill_(0xFF)
        return t
    try:
        torch.empty = poisoned_empty
        got = run()
    finally:
        torch.empty = orig_empty
    for a, b in zip(ref, got):
        assert torch.equal(a.cpu(), b.cpu())


def test_idr_vertex_colour_mid_network():
    """plain NeuS (idr colour mode: view_dirs = -gradients, PE-4) with a skip connection."""
    fx = G.load("functions")
    P = G.prefixed(fx, "midw:")
    ocfg = G.mid_config()
    r = N.make_renderer(ocfg, P, None, DEV)
    from oracle import colorneus_oracle as O
    pts = torch.from_numpy(fx["mid:pts"])
    sdf, feat, g = O.sdf_forward(P, ocfg.sdf, pts, want_grad=True)
    want = O.color_forward(P, ocfg.color, pts, g, -g, feat)
    rgb = r.extract_color(fx["mid:pts"], DEV)
    assert G.relerr(rgb, want.detach()) < TOL
    assert G.relerr(r.sdf(pts.to(DEV)).cpu()[:, 0], fx["mid:sdf_out"][:, 0]) < TOL


def test_full_size_properties_4096_rays():
    """BASELINE-size batch (4096 rays x 128 samples, DTU network, trained-like weights): size-independent properties.
    (1) compositing invariants of the outputs, (2) ray additivity: with a ray-separable objective the parameter gradients of
    the whole batch equal the sum of the gradients of its two halves (what ray-sharding relies on), (3) run-to-run determinism."""
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
        # ray-separable objective: plain sums over rays (no means over the batch, no ratio of batch sums)
        obj = ((out["color_fine"] - gt[idx]) ** 2).sum() + 0.1 * out["weight_sum"].sum() + 0.01 * (out["gradients"] ** 2).sum() \
            + 0.05 * out["delta_relight"].sum()
        for p in r.parameters():
            p.grad = None
        obj.backward()
        return out, {k: p.grad.clone() for k, p in r.named_parameters()}

    allr = torch.arange(4096, device=DEV)
    out, g_all = run(allr)
    # (1) invariants
    w = out["weights"].detach()
    assert bool((w >= 0).all()) and bool(torch.isfinite(w).all())
    assert float((w.sum(-1, keepdim=True) - out["weight_sum"].detach()).abs().max()) < 1e-5
    assert float(out["weight_sum"].max()) <= 1.0 + 1e-5
    assert bool((out["weight_max"].reshape(-1) <= w.max(-1).values + 1e-7).all())
    z = out["z_vals"]
    assert bool((z[:, 1:] >= z[:, :-1]).all()), "samples must be sorted along every ray"
    ins = out["inside_sphere"]
    assert bool(((ins == 0) | (ins == 1)).all())
    assert bool((out["color_fine"] >= -1e-6).all()) and bool((out["color_fine"] <= 1 + 1e-5).all())
    # (2) additivity over rays
    _, g_a = run(allr[:2048])
    _, g_b = run(allr[2048:])
    for k, g in g_all.items():
        err = float((g - (g_a[k] + g_b[k])).abs().max())
        # own scale per tensor; measured 2.1e-6 at worst (tools/gate_probe.py: only the summation order over the point ranges and the power-of-two
        # exponent of the weight-gradient accumulators differ between the whole batch and its halves)
        assert err <= 1e-5 * float(g.abs().max()), (k, err, float(g.abs().max()))
    # (3) determinism
    _, g_again = run(allr)
    for k, g in g_all.items():
        assert torch.equal(g, g_again[k]), k


_FUSED_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import _native as N
from oracle import colorneus_oracle as O
ocfg = O.dtu_config()
P = O.init_params(ocfg, seed=0, dtype=torch.float32, trained_like=True)
r = N.make_renderer(ocfg, P, None, "cuda:0")
g = torch.Generator().manual_seed(17)
out = {}
for n in (100, 20001, 70003):
    pts = (torch.rand(n, 3, generator=g) * 2 - 1).to("cuda:0")
    out["sdf%d" % n] = r.sdf(pts).cpu().numpy()
np.savez(sys.argv[2], **out)
"""


def test_fused_sdf_chain_matches_per_layer_kernels(tmp_path):
    """The chain-fused SDF value kernel (cnr_chain.hip; three tile heights, ragged last tiles) against the per-layer kernels
    (CNR_NO_FUSED=1, child process: the switch is read once per process) on the DTU-size network, and against the float64 oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("fused", {}), ("layers", {"CNR_NO_FUSED": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _FUSED_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    from oracle import colorneus_oracle as O
    ocfg = O.dtu_config()
    P64 = {k: v.double() for k, v in O.init_params(ocfg, seed=0, dtype=torch.float32, trained_like=True).items()}
    g = torch.Generator().manual_seed(17)
    for n in (100, 20001, 70003):
        a, b = res["fused"]["sdf%d" % n], res["layers"]["sdf%d" % n]
        scale = float(np.abs(b).max())
        assert float(np.abs(a - b).max()) < 2e-6 * scale, n     # same split / scales / MFMA order; only the narrow top layer's summation order differs
        pts = torch.rand(n, 3, generator=g) * 2 - 1
        if n <= 20001:
            ref = O.sdf_value(P64, ocfg.sdf, pts.double()).numpy().reshape(-1)
            assert float(np.abs(a.reshape(-1) - ref).max()) < 2e-5 * scale, n


_CHAIN_CHILD = r"""
import sys, os
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import _native as N
import color_neus_amd as cn
from oracle import colorneus_oracle as O
res = {}
for tag, (ns, ni, R) in {"rt1": (24, 24, 37), "rt2": (40, 40, 300), "rt4": (40, 40, 500)}.items():   # P = 1776 / 24000 / 40000 points: ragged last tiles at every tile height
    ocfg = O.dtu_config(n_samples=ns, n_importance=ni)
    P = O.init_params(ocfg, seed=5, trained_like=True)
    r = N.make_renderer(ocfg, P, None, "cuda:0")
    g = torch.Generator().manual_seed(13)
    o = torch.randn(R, 3, generator=g); o = o / o.norm(dim=-1, keepdim=True) * 2.7
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.3 - o, dim=-1)
    near, far = O.near_far_from_sphere(o, d)
    gt, mask = torch.rand(R, 3, generator=g), (torch.rand(R, generator=g) > 0.3).float()
    outp = r(o.cuda(), d.cuda(), near.cuda(), far.cuda(), perturb_overwrite=0)
    loss, _ = cn.compute_loss(outp, gt.cuda(), mask.cuda())
    loss.backward()
    res.update({tag + ":out:" + k: v.detach().cpu().numpy() for k, v in outp.items() if torch.is_tensor(v)})
    res.update({tag + ":g:" + k: p.grad.detach().cpu().numpy() for k, p in r.named_parameters()})
np.savez(out, **res)
"""


def test_chain_fused_forward_matches_per_layer_kernels(tmp_path):
    """The chain-fused SAVING forward kernels (cnr_chain_fwd.hip: SDF network, colour + relight stacks) against the per-layer launches
    (CNR_NO_CHAIN_FWD=1 CNR_NO_CHAIN_SDF=1, child processes) at the three tile heights, each with a ragged last tile (points per ray not a multiple of the tile).
    Same products in the same order; the epilogues round once where the per-layer kernels round twice and the heads sum in another order, so
    the sdf differs in its last bits, which the logistic CDF at the trained-like sharpness carries into the weights: outputs agree to 1e-5 of
    their scale at the median entry and 1e-4 at the worst (measured 2.2e-5).  Gradients: a ReLU unit within that round-off of zero may fall on the other side in one build
    (see tests/_golden.py), which moves that point's contribution: 2e-5 at the median entry of every tensor, 2e-2 at the worst."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("fused", {}), ("layers", {"CNR_NO_CHAIN_FWD": "1", "CNR_NO_CHAIN_SDF": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _CHAIN_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    assert set(res["fused"]) == set(res["layers"])
    bad = []
    for k in sorted(res["fused"]):
        a, b = res["fused"][k].astype(np.float64), res["layers"][k].astype(np.float64)
        den = max(float(np.abs(b).max()), 1e-300)
        e = np.abs(a - b).reshape(-1) / den
        if ":out:" in k:
            if k.endswith("z_vals"):
                ok = np.array_equal(a, b)                                           # the sampler's value chains are the same in both builds
            else:
                ok = float(np.median(e)) < 1e-5 and float(e.max()) < 1e-4
        else:
            ok = float(np.median(e)) < 2e-5 and float(e.max()) < 2e-2
        if not ok:
            bad.append((k, float(np.median(e)), float(e.max())))
    assert not bad, bad


_STREAM_CHILD = r"""
import sys, os
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import _native as N
fx, r, o_, loss, grads, o, d = N.run_native("dtu_sharp", "det", None, torch.device("cuda:0"), fixed_z=True)
res = {"out:" + k: v.detach().cpu().numpy() for k, v in o_.items() if torch.is_tensor(v)}
res.update({"g:" + k: v.detach().cpu().numpy() for k, v in grads.items()})
res["d_o"] = o.grad.cpu().numpy(); res["d_d"] = d.grad.cpu().numpy()
np.savez(out, **res)
"""


def test_stream_form_of_the_layer_kernel_matches_the_general_form(tmp_path):
    """layer_gemm_ws_stream_kernel (per-wave-group loops, prefetched epilogue inputs; most layer launches of a step) against the general
    kernel (CNR_WS_NOSTREAM=1, child processes): same tiles, same split, same epilogue arithmetic.  Since round 6 the stream form issues its
    products as v_mfma_f32_16x16x32_f16 and the general form stays on 32x32x16 (cnr_gemm_ws.h, WS_GEN_MFMA16: the general form is 30 % slower on the
    other shape), so the fp32 accumulation order inside a product differs: every output and every gradient of a forward + backward pass must
    agree to fp32 round-off -- 1e-5 of the tensor's largest entry for the outputs, the fixture's own per-tensor float32 tolerance for the gradients
    (a build with -DWS_GEN_MFMA16=1 is bit-identical: that was this test until then)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("stream", {}), ("general", {"CNR_WS_NOSTREAM": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _STREAM_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    assert set(res["stream"]) == set(res["general"])
    fx = G.load("dtu_sharp")
    bad, worst = [], {}
    frac, floor, cap = G._gate(True)
    for k in sorted(res["stream"]):
        a, b = res["stream"][k].astype(np.float64).reshape(-1), res["general"][k].astype(np.float64).reshape(-1)
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        if a.size == 0:
            continue
        e = np.nan_to_num(np.abs(a - b)) / max(float(np.nanmax(np.abs(b))), 1e-300)
        grad = k.startswith("g:") or k in ("d_o", "d_d")
        if not grad:
            lim, emax, bulk = 1e-5, float(e.max()), float(e.max())
        else:
            # Two float32 evaluations of the same gradient under the HIP gate's own rule (G.check_param_grads, strict): the bulk of a tensor's entries
            # within the fixture's per-tensor float32 tolerance (`gspread`; 5e-5 for the ray gradients), a handful -- a ReLU unit of the colour / relight
            # stacks that sits within round-off of its kink at one sample and flips with the last bit of the normals -- up to the hard cap
            sp = fx.get("det:gspread:" + k[2:]) if k.startswith("g:") else None
            lim = 5e-5 if sp is None else min(cap, G.grad_tolerance(sp, strict=True) if a.size > 1 else G.scalar_tolerance(sp))
            allowed = max(floor, int(frac * a.size)) if a.size > 1 else 0
            emax = float(e.max())
            bulk = float(np.sort(e)[-(allowed + 1)]) if a.size > allowed else 0.0
            if emax > cap:
                bad.append((k, "cap", emax, cap))
        worst["grad" if grad else "out"] = max(worst.get("grad" if grad else "out", 0.0), bulk / lim)
        if not bulk < lim:
            bad.append((k, bulk, lim, emax))
    print("stream vs general form, largest difference as a fraction of the limit:", worst)
    assert not bad, bad


def test_fused_layer_dw_matches_separate_launches(tmp_path):
    """layer_dw_kernel (cnr_gemm_fdw.hip: a backward layer launch that also forms the weight gradient of its layer on chip) against the
    separate layer + weight-gradient launches (CNR_NO_FDW=1, child processes): outputs are the same arithmetic and must agree to the bit;
    weight gradients differ only in the summation order over points and the power-of-two exponent of the split, and the ray gradients only
    through the top SDF layer, whose fused value-backward launch takes the sdf column of the cotangent as an fp32 rank-one update instead of a
    257th contraction index of the split-f16 product (round-off of one term)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("fused", {}), ("separate", {"CNR_NO_FDW": "1"}), ("top_unfused", {"CNR_NO_TOP_FUSE": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _STREAM_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    bad = []
    for other in ("separate", "top_unfused"):   # (top_unfused: only the top SDF layer's backward as launches of its own, everything else fused)
        assert set(res["fused"]) == set(res[other])
        for k in sorted(res["fused"]):
            a, b = res["fused"][k].astype(np.float64), res[other][k].astype(np.float64)
            if k.startswith("g:") or k in ("d_o", "d_d"):
                e = float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-300)
                if not e < 5e-6:
                    bad.append((other, k, e))
            elif not np.array_equal(a, b, equal_nan=True):
                bad.append((other, k, "not bit-identical"))
    assert not bad, bad


def test_streaming_strip_and_head_backward_match_gemm_launches(tmp_path):
    """strip_bwd_kernel (the input columns beyond 256 of the relight y-layer and of colour layer 0) and head_bwd_kernel (the 3-wide heads) are
    streaming restatements of what the narrow layer GEMM + weight-gradient strip launches computed (CNR_NO_STRIP_BWD=1 / CNR_NO_HEAD_BWD=1,
    child processes): same fp32 products in another summation order, so every output and gradient agrees to round-off of its own scale."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("stream", {}), ("gemm", {"CNR_NO_STRIP_BWD": "1", "CNR_NO_HEAD_BWD": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _STREAM_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    assert set(res["stream"]) == set(res["gemm"])
    bad = []
    for k in sorted(res["stream"]):
        a, b = res["stream"][k].astype(np.float64), res["gemm"][k].astype(np.float64)
        e = float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-300)
        if not e < 5e-6:
            bad.append((k, e))
    assert not bad, bad


def test_narrow_input_layer_kernels_match_separate_launches(tmp_path):
    """sweep0_dw_kernel (cnr_sweep0.hip: sweep launch of the first SDF layer + its gradient-chain weight-gradient pair) and narrow_bwd_kernel
    (cnr_narrow_bwd.hip: input cotangent + weight + bias gradient of the relight in_layer and of the first SDF layer in one pass) against the
    separate layer / weight-gradient launches (CNR_NO_SWEEP0=1 CNR_NO_NARROW_BWD=1, child processes).  The sweep launch's own outputs are the same
    arithmetic; the weight gradients and the input cotangents change from FP32-MFMA products to split-f16 products (both at fp32 round-off of
    the result), so outputs agree to the bit and gradients to round-off of their own scale."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (CNR_NO_NARROW_DX=1 in both runs: the forward use of the same kernel -- the 39-column end of the gradient chain -- changes the normals at
    # round-off level and with them every output; it is held to the oracle by the strict gates and to its fallback by test_fallback_kernels_keep_parity)
    both = {"CNR_NO_NARROW_DX": "1"}
    for tag, extra in (("fused", both), ("separate", dict(both, CNR_NO_SWEEP0="1", CNR_NO_NARROW_BWD="1"))):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _STREAM_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    # ... and the three batches of _CHAIN_CHILD: 1776 / 24000 / 40000 points, i.e. a ragged last 32-point tile and point ranges of 1 / 3 / 5 tiles
    for tag, extra in (("fused", both), ("separate", dict(both, CNR_NO_SWEEP0="1", CNR_NO_NARROW_BWD="1"))):
        path = str(tmp_path / (tag + "_ragged.npz"))
        r = subprocess.run([sys.executable, "-c", _CHAIN_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag].update(dict(np.load(path)))
    assert set(res["fused"]) == set(res["separate"])
    bad = []
    for k in sorted(res["fused"]):
        a, b = res["fused"][k].astype(np.float64), res["separate"][k].astype(np.float64)
        if k.startswith("g:") or ":g:" in k or k in ("d_o", "d_d"):
            e = float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-300)
            if not e < 5e-6:
                bad.append((k, e))
        elif not np.array_equal(a, b, equal_nan=True):
            bad.append((k, "differs"))
    assert not bad, bad


_SAMPLER_CHILD = r"""
import sys, os
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import _native as N
res = {}
for name in ("dtu_sharp", "tiny_sharp"):
    fx, r, o_, loss, grads, o, d = N.run_native(name, "det", None, torch.device("cuda:0"), fixed_z=False)
    res.update({name + ":out:" + k: v.detach().cpu().numpy() for k, v in o_.items() if torch.is_tensor(v)})
    res.update({name + ":g:" + k: v.detach().cpu().numpy() for k, v in grads.items()})
np.savez(out, **res)
"""


def test_fused_sampler_step_is_bit_identical(tmp_path):
    """sampler_step_kernel (merge of the previous iteration + up_sample + embedding of the new samples in one launch per iteration) against
    the three launches it replaces (CNR_NO_SAMPLER_FUSE=1, child processes): same arithmetic on the same values -> z_vals, every output and
    every gradient agree to the bit."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("fused", {}), ("separate", {"CNR_NO_SAMPLER_FUSE": "1"})):
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _SAMPLER_CHILD, root, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    assert set(res["fused"]) == set(res["separate"]) and any(k.endswith("out:z_vals") for k in res["fused"])
    for k in sorted(res["fused"]):
        assert np.array_equal(res["fused"][k], res["separate"][k], equal_nan=True), k
