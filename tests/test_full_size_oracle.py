"""GPU (MI355X): BASELINE config C3 at FULL size against the float64 oracle run live on the host.

1024 rays x 128 samples (64 + 64 in 4 up-sampling steps: the renderer block of Color_NeuS_bmvs.yml = the DTU block), trained-like
weights (inv_s = 665), one training loss.  At this size the fused layer + weight-gradient launches run with 32 tiles per point range
(the 16-ray fixtures: 1; the 160-ray test: 4-5), the running-minimum exponent of the weight-gradient accumulators is rescaled many
times and every partial-sum slot of the pool is in use -- none of which the fixture-size comparisons exercise.
Compared under the strict rule of tests/_golden.py (check_grads_full): the 12 outputs, the loss, all 53 parameter gradients (EVERY
entry) and d rays_o / d rays_d, each at its own scale against float64; the oracle's own float32 run calibrates the per-tensor tolerance.
The table goes to gpurun_out/param_grad_error_table_c3.txt (committed under profiles/)."""
import os
import time

import pytest
import torch

import _golden as G
import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(R, seed):
    from oracle import colorneus_oracle as O
    from color_neus_amd import synthetic
    ocfg = O.dtu_config()
    P = O.init_params(ocfg, seed=5, trained_like=True)
    views = synthetic.synthetic_view(seed=seed, device="cpu")
    sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(seed + 100))[:R]
    o, d, near, far, gt, mask = [x[sel].contiguous() for x in views]
    t_rand = torch.rand(R, 1, generator=torch.Generator().manual_seed(seed + 200))
    return O, ocfg, P, o, d, near, far, t_rand, gt, mask


def _oracle_run(O, ocfg, P, o, d, near, far, z, gt, mask, dt):
    Pd = {k: v.to(dt).clone().requires_grad_(True) for k, v in P.items()}
    od, dd = o.to(dt).clone().requires_grad_(True), d.to(dt).clone().requires_grad_(True)
    out = O.render(Pd, ocfg, od, dd, near.to(dt), far.to(dt), z_vals=z.to(dt))
    loss, _ = O.compute_loss(out, gt.to(dt), mask.to(dt))
    loss.backward()
    g = {k: v.grad for k, v in Pd.items()}
    g["rays_o"], g["rays_d"] = od.grad, dd.grad
    return {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}, loss.detach(), g


def test_c3_full_size_against_live_float64_oracle():
    import color_neus_amd as cn
    R = 1024
    O, ocfg, P, o, d, near, far, t_rand, gt, mask = _batch(R, seed=31)
    r = N.make_renderer(ocfg, P, None, DEV)
    # G1 at full size: the HIP sampler against the oracle's float32 sampler on the same jitter draw
    orig = torch.rand
    try:
        torch.rand = lambda *a, **k: t_rand.clone()
        with torch.no_grad():
            z_hip = r(o.to(DEV), d.to(DEV), near.to(DEV), far.to(DEV))["z_vals"].cpu()
    finally:
        torch.rand = orig
    z32 = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    # The sampler amplifies last-bit differences: on a handful of rays a searchsorted / denom < 1e-5 decision flips and the
    # samples of one up-sampling step move by a fraction of a coarse section.  The band is measured here on the same batch from the oracle
    # itself (float32 against float64: 5-7 rays of these 1024 beyond 1e-3, largest deviation 0.0154); the HIP sampler must stay inside it:
    # at most max(6, 2 x band) rays beyond 1e-3 (measured: 4), no sample further off than HALF a coarse section (1 / N_SAMPLES = 0.0156;
    # measured 0.0071).
    P64 = {k: v.double() for k, v in P.items()}
    z64 = O.sample_z(P64, ocfg, o.double(), d.double(), near.double(), far.double(), t_rand.double())
    band = int(((z32.double() - z64).abs().max(1).values > 1e-3).sum())
    dz = (z_hip - z32).abs()
    moved = int((dz.max(1).values > 1e-3).sum())
    assert moved <= max(6, 2 * band), (moved, band)
    assert float(dz.max()) <= 1.0 / ocfg.n_samples, float(dz.max())
    assert bool((z_hip[:, 1:] >= z_hip[:, :-1]).all())
    # G2 at full size, identical z
    t0 = time.time()
    out64, l64, g64 = _oracle_run(O, ocfg, P, o, d, near, far, z32, gt, mask, torch.float64)
    out32, l32, g32 = _oracle_run(O, ocfg, P, o, d, near, far, z32, gt, mask, torch.float32)
    t_oracle = time.time() - t0
    og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    out = r(og, dg, near.to(DEV), far.to(DEV), z_vals=z32.to(DEV))
    loss, _ = cn.compute_loss(out, gt.to(DEV), mask.to(DEV))
    loss.backward()
    lines = ["# C3 size (1024 rays x 128 samples, DTU / BlendedMVS renderer block, trained-like weights): HIP vs the float64 oracle at identical z",
             "# G1 at this size: %d rays with a sample more than 1e-3 from the float32 oracle's (largest %.4f); the oracle's own float32-vs-float64 band: %d rays" % (moved, float(dz.max()), band),
             "# oracle float64 + float32 runs on the host: %.1f s" % t_oracle,
             "%-44s %10s %12s %12s" % ("output", "numel", "err_vs_f64", "f32_oracle")]
    bad = []
    # per-sample weights / cdf at inv_s = 665 are not reproducible to 1e-4 in float32 by any implementation (see
    # test_hip_parity.test_against_oracle_larger_batch): their tolerance is 3 x the float32 oracle's own distance from float64, capped at 5e-4
    for k in G.OUTPUT_KEYS:
        e = G.relerr(out[k].detach().cpu().reshape(out64[k].shape), out64[k])
        e32 = G.relerr(out32[k].double(), out64[k])
        lim = min(5e-4, max(1e-4, 3.0 * e32)) if k in ("weights", "weight_max", "cdf_fine") else 1e-4
        lines.append("%-44s %10d %12.2e %12.2e" % (k, out64[k].numel(), e, e32))
        if not e < lim:
            bad.append((k, e, lim))
    el = abs(float(loss.detach()) - float(l64)) / abs(float(l64))
    lines.append("%-44s %10d %12.2e %12.2e" % ("loss", 1, el, abs(float(l32) - float(l64)) / abs(float(l64))))
    if not el < 1e-4:
        bad.append(("loss", el, 1e-4))
    got = {(k[len("renderer."):] if k.startswith("renderer.") else k): p.grad for k, p in r.named_parameters()}
    got["rays_o"], got["rays_d"] = og.grad, dg.grad
    assert set(got) == set(g64), set(got) ^ set(g64)
    lines.append("%-44s %10s %12s %12s %12s %10s %8s" % ("gradient", "numel", "err_max", "err_bulk(1%)", "f32_oracle", "tol", "n>tol"))
    cap = G.STRICT_TOL_CAP
    for k, r64 in g64.items():
        r64 = r64.double().reshape(-1)
        den = max(float(r64.abs().max()), 1e-300)
        e = (got[k].detach().cpu().double().reshape(-1) - r64).abs() / den
        spread = float((g32[k].double().reshape(-1) - r64).abs().max()) / den
        lim = min(cap, G.grad_tolerance(spread, True))
        allowed = G._allowed(e.numel(), True)
        bulk = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        lines.append("%-44s %10d %12.2e %12.2e %12.2e %10.1e %8d" % (k, e.numel(), float(e.max()), bulk, spread, lim, int((e > lim).sum())))
    # strict rule + at this size the LARGEST error of every tensor within max(1e-4, 1.5 x the float32 oracle's largest error on it)
    bad += G.check_grads_full(g64, g32, got, strict=True, rel_max=G.STRICT_SPREAD_FACTOR)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "param_grad_error_table_c3.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert not bad, bad


def _separable_loss(out, gt, mask, R_total, M):
    """The training objective (NeuS_Trainer.py:129-171) with its two non-separable terms replaced by ray-separable stand-ins of the same
    shape and size, normalised by the counts of the WHOLE batch, so that the value and every gradient are sums over rays:
    eikonal: sum inside * (|g| - 1)^2 / (R M) (the reference divides by the global sum of its relax mask, Color_NeuS.py:122-123);
    relight: 0.02 * sum(delta_relight * mask) / (3 R M) (the reference squares the global mean, NeuS_Trainer.py:153: gradient 2 * mean * d mean)."""
    rgb = ((out["color_fine"] - gt) ** 2).sum() / (3.0 * R_total)
    ws = out["weight_sum"].reshape(-1).clip(1e-3, 1.0 - 1e-3)
    bce = -(mask * torch.log(ws) + (1.0 - mask) * torch.log(1.0 - ws)).sum() / R_total
    gn = out["gradients"].norm(dim=-1)
    eik = (out["inside_sphere"].detach() * (gn - 1.0) ** 2).sum() / (R_total * M)
    rel = (out["delta_relight"] * mask[:, None, None]).sum() / (3.0 * R_total * M)
    return rgb + 0.1 * eik + 0.1 * bce + 0.02 * rel


def test_c4_4096_rays_ray_separable_objective_against_float64_oracle_in_quarters():
    """BASELINE config C4's batch (4096 rays x 128 samples) on ONE GPU against float64: with the ray-separable form of the training objective
    above the float64 oracle can evaluate the batch in four 1024-ray quarters (the host-memory footprint of the C3 test) whose parameter
    gradients add up to the gradient of the whole batch -- the native path runs the 4096 rays as ONE call (128 tiles per point range in the
    fused layer + weight-gradient launches, the full partial-sum pool).  Every entry of all 53 parameter gradients and d rays under the
    strict rule, the largest error of each tensor within max(1e-4, 1.5 x the float32 oracle's)."""
    R, Q = 4096, 4
    O, ocfg, P, o, d, near, far, t_rand, gt, mask = _batch(R, seed=41)
    M = ocfg.n_samples + ocfg.n_importance
    t0 = time.time()
    z32 = torch.cat([O.sample_z(P, ocfg, o[a:a + R // Q], d[a:a + R // Q], near[a:a + R // Q], far[a:a + R // Q], t_rand[a:a + R // Q]) for a in range(0, R, R // Q)])
    tot, vals = {}, {}
    for dt in (torch.float64, torch.float32):
        acc, dro, drd, val = None, [], [], 0.0
        for a in range(0, R, R // Q):
            sl = slice(a, a + R // Q)
            Pd = {k: v.to(dt).clone().requires_grad_(True) for k, v in P.items()}
            od, dd = o[sl].to(dt).clone().requires_grad_(True), d[sl].to(dt).clone().requires_grad_(True)
            out = O.render(Pd, ocfg, od, dd, near[sl].to(dt), far[sl].to(dt), z_vals=z32[sl].to(dt))
            L = _separable_loss(out, gt[sl].to(dt), mask[sl].to(dt), R, M)
            L.backward()
            val += float(L.detach())
            gq = {k: v.grad.detach().double() for k, v in Pd.items()}
            acc = gq if acc is None else {k: acc[k] + gq[k] for k in acc}
            dro.append(od.grad.detach().double()); drd.append(dd.grad.detach().double())
            del out, L, Pd
        acc["rays_o"], acc["rays_d"] = torch.cat(dro), torch.cat(drd)
        tot[dt], vals[dt] = acc, val
    t_oracle = time.time() - t0
    r = N.make_renderer(ocfg, P, None, DEV)
    og, dg = o.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    out = r(og, dg, near.to(DEV), far.to(DEV), z_vals=z32.to(DEV))
    L = _separable_loss(out, gt.to(DEV), mask.to(DEV), R, M)
    L.backward()
    assert abs(float(L.detach()) - vals[torch.float64]) < 1e-4 * abs(vals[torch.float64]), (float(L.detach()), vals)
    got = {(k[len("renderer."):] if k.startswith("renderer.") else k): p.grad for k, p in r.named_parameters()}
    got["rays_o"], got["rays_d"] = og.grad, dg.grad
    g64, g32 = tot[torch.float64], tot[torch.float32]
    lines = ["# C4 batch on one GPU (4096 rays x 128 samples as ONE call) vs the float64 oracle in four 1024-ray quarters, ray-separable form of the training objective",
             "# oracle float64 + float32 runs on the host: %.1f s; objective %.6f (float64 %.6f)" % (t_oracle, float(L.detach()), vals[torch.float64]),
             "%-44s %10s %12s %12s %12s" % ("gradient", "numel", "err_max", "err_bulk(1%)", "f32_oracle")]
    for k, r64 in g64.items():
        r64 = r64.reshape(-1)
        den = max(float(r64.abs().max()), 1e-300)
        e = (got[k].detach().cpu().double().reshape(-1) - r64).abs() / den
        spread = float((g32[k].reshape(-1) - r64).abs().max()) / den
        allowed = G._allowed(e.numel(), True)
        bulk = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        lines.append("%-44s %10d %12.2e %12.2e %12.2e" % (k, e.numel(), float(e.max()), bulk, spread))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "param_grad_error_table_c4.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    bad = G.check_grads_full(g64, g32, got, strict=True, rel_max=G.STRICT_SPREAD_FACTOR)
    assert not bad, bad
