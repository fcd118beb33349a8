"""Helpers shared by the oracle (CPU) and HIP (GPU) parity tests: fixture loading."""
import os

import numpy as np
import torch

from oracle import colorneus_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "tiny_init": lambda: O.tiny_config(),
    "tiny_sharp": lambda: O.tiny_config(),
    "tiny_sharp_anneal": lambda: O.tiny_config(),       # cos_anneal_ratio 0.3 + background_rgb (the fixture's call:* entries)
    "dtu_sharp_anneal": lambda: O.dtu_config(),
    "tiny_noimp_sharp": lambda: _noimp(),
    "tiny_neus_sharp": lambda: O.RenderConfig(
        type="NeuS", n_samples=16, n_importance=16,
        sdf=O.SDFConfig(d_out=65, d_hidden=64, n_layers=2, skip_in=[]),
        color=O.ColorConfig(d_feature=64, mode="idr", d_in=9, d_hidden=64, n_layers=2, multires_view=4), relight=None),
    "tiny_outside": lambda: _outside(O.tiny_config()),
    "tiny_neus_outside": lambda: _outside(CONFIGS["tiny_neus_sharp"]()),
    "dtu_init": lambda: O.dtu_config(),
    "dtu_sharp": lambda: O.dtu_config(),
    "dtu_noimp_sharp": lambda: _noimp_dtu(),
    "neus_dtu_sharp": lambda: O.RenderConfig(type="NeuS", relight=None),
}


# Round 6: the network-configuration branches the ABI accepts but no shipped YAML takes (fields.py:72-73 WEIGHT_NORM, :170-171 MODE no_normal,
# :186-187 SQUEEZE_OUT, :302-306 / :342-343 INCLUDE_GRAD, :311-314 / :346-349 Y_IN_LAYER incl. the rgb-into-the-last-layer form, :354-359 INV_SIGMOID,
# :45-48 SKIP_IN at another layer and TWO skip connections), each at the tiny size (per-layer kernels) and at the DTU widths (chain-fused kernels).  tools/gen_golden.py captures
# them from the reference with these very configurations.  The tiny ones also move the remaining cnr_config fields off their defaults:
# UP_SAMPLE_STEPS 2, SDF MULTIRES 4 and SCALE 2, MULTIRES_VIEW 2 (relight; colour in no_normal mode).
def _tiny(**kw):
    c = O.tiny_config()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


VARIANTS = {
    "tiny_rel_alt": lambda: _tiny(relight=O.RelightConfig(d_hidden=64, n_layers=2, y_in_layer=2, include_grad=False, inv_sigmoid=False, multires_view=2)),
    "tiny_nown_skip2": lambda: _tiny(
        up_sample_steps=2,
        sdf=O.SDFConfig(d_out=65, d_hidden=64, n_layers=5, skip_in=[2], weight_norm=False, multires=4, scale=2.0),
        color=O.ColorConfig(d_feature=64, mode="no_view_dir", d_in=6, d_hidden=64, n_layers=2, multires_view=0, weight_norm=False),
        relight=O.RelightConfig(d_hidden=64, n_layers=3, y_in_layer=2)),
    "tiny_twoskip": lambda: _tiny(sdf=O.SDFConfig(d_out=65, d_hidden=64, n_layers=5, skip_in=[2, 4])),
    "tiny_nosq": lambda: _tiny(color=O.ColorConfig(d_feature=64, mode="no_view_dir", d_in=6, d_hidden=64, n_layers=2, multires_view=0, squeeze_out=False)),
    "tiny_neus_nonormal": lambda: O.RenderConfig(
        type="NeuS", n_samples=16, n_importance=16,
        sdf=O.SDFConfig(d_out=65, d_hidden=64, n_layers=2, skip_in=[]),
        color=O.ColorConfig(d_feature=64, mode="no_normal", d_in=6, d_hidden=64, n_layers=2, multires_view=2, squeeze_out=False), relight=None),
    "dtu_rel_alt": lambda: O.RenderConfig(
        type="Color_NeuS", color=O.ColorConfig(mode="no_view_dir", d_in=6, multires_view=0),
        relight=O.RelightConfig(y_in_layer=2, include_grad=False, inv_sigmoid=False)),
    "dtu_nown_skip6": lambda: O.RenderConfig(
        type="Color_NeuS", sdf=O.SDFConfig(skip_in=[6], weight_norm=False),
        color=O.ColorConfig(mode="no_view_dir", d_in=6, multires_view=0, weight_norm=False),
        relight=O.RelightConfig(y_in_layer=4)),
    "dtu_twoskip": lambda: O.RenderConfig(type="Color_NeuS", sdf=O.SDFConfig(skip_in=[3, 6]), color=O.ColorConfig(mode="no_view_dir", d_in=6, multires_view=0)),
    "neus_dtu_nonormal": lambda: O.RenderConfig(type="NeuS", color=O.ColorConfig(mode="no_normal", d_in=6, squeeze_out=False), relight=None),
}
CONFIGS.update(VARIANTS)


def _outside(c, n=8):
    c.n_outside = n
    return c


def _noimp():
    c = O.tiny_config()
    c.n_importance = 0
    return c


def _noimp_dtu():
    c = O.dtu_config()
    c.n_importance = 0
    return c


def mid_config():
    return O.RenderConfig(type="NeuS", n_samples=16, n_importance=16,
                          sdf=O.SDFConfig(d_out=65, d_hidden=64, n_layers=6, skip_in=[3]),
                          color=O.ColorConfig(d_feature=64, mode="idr", d_in=9, d_hidden=64, n_layers=3, multires_view=4),
                          relight=None)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def weights_of(name, fx, dtype=torch.float32):
    """Weights of an e2e fixture: stored (tiny) or regenerated from the recipe (checksum verified)."""
    cfg = CONFIGS[name]()
    stored = {k[2:]: torch.from_numpy(v).to(dtype) for k, v in fx.items() if k.startswith("w:")}
    if stored:
        if "nerf_seed" in fx:   # NeRF++ background weights: recipe + checksum
            nerf = O.init_nerf_params(seed=int(fx["nerf_seed"]))
            cs = O.params_checksum(nerf)
            assert abs(cs - float(fx["nerf_checksum"])) <= 1e-9 * abs(cs), "nerf weight recipe drifted from the fixture"
            stored.update({k: v.to(dtype) for k, v in nerf.items()})
        return cfg, stored
    P = O.init_params(cfg, seed=int(fx["weight_seed"]), dtype=torch.float32, trained_like=bool(fx["trained_like"]))
    cs = O.params_checksum(P)
    assert abs(cs - float(fx["weight_checksum"])) <= 1e-9 * abs(cs), "weight recipe drifted from the fixture"
    return cfg, {k: v.to(dtype) for k, v in P.items()}


def call_kwargs(fx, device=None, dtype=torch.float32):
    """Non-default call arguments a fixture was captured with (tools/gen_golden.py: cos_anneal_ratio, background_rgb as the [1, 3] tensor of
    Color_NeuS.py:104-106); empty for the default-call fixtures."""
    if "call:cos_anneal_ratio" not in fx:
        return {}
    bg = torch.from_numpy(np.asarray(fx["call:background_rgb"])).to(dtype).reshape(1, 3)
    return dict(cos_anneal_ratio=float(fx["call:cos_anneal_ratio"]), background_rgb=bg.to(device) if device is not None else bg)


def prefixed(fx, prefix, dtype=torch.float32):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in fx.items() if k.startswith(prefix)}


def relerr(a, b):
    """max-abs error normalised by max-abs of the reference value (SURVEY 8c convention)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    den = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / den


def check_g1(z, fx, tag):
    """Gate G1 (sampler): every sample position within 1e-3 of the reference's.  The up-sampling steps amplify float32 round-off (the
    reference's own float32 and float64 runs place samples up to 6.7e-4 apart on one draw, 3e-3 on another), so a ray on which THE REFERENCE'S OWN
    float32 and float64 samplers disagree by more than 2e-4 is a ray on which any float32 implementation may land elsewhere: where the fixture holds
    that float64 run (round-6 fixtures: `f64:z_vals_own`) those rays may sit up to HALF A COARSE SECTION at 64 samples (1 / 64: a new sample that falls into
    the neighbouring section of the piecewise-linear cdf; the full-size test uses the same bound) away, every other ray must be within 1e-3; the older fixtures keep the flat rule with one ray per 512 (at least one) allowed up to 4e-3.  None when fine."""
    ref = torch.from_numpy(np.asarray(fx[f"{tag}:z_vals"])).double()
    df = (torch.as_tensor(z).double().cpu() - ref).abs()
    bad = (df > 1e-3).any(dim=1)
    if f"{tag}:f64:z_vals_own" in fx:
        own = (torch.from_numpy(np.asarray(fx[f"{tag}:f64:z_vals_own"])).double() - ref).abs().amax(dim=1) > 2e-4
        stray = int((bad & ~own).sum())
        if float(df.max()) >= 1.0 / 64 or stray > 0:
            return "z_vals: max |dz| %.2e; %d ray(s) beyond 1e-3 that the reference's own float32 / float64 samplers agree on (%d sensitive rays)" % (float(df.max()), stray, int(own.sum()))
        return None
    allowed = max(1, df.shape[0] // 512)
    if float(df.max()) >= 4e-3 or int(bad.sum()) > allowed:
        return "z_vals: max |dz| %.2e, %d ray(s) beyond 1e-3 (allowed %d below 4e-3)" % (float(df.max()), int(bad.sum()), allowed)
    return None


def check_outputs(fx, tag, out, tol=1e-4, factor=1.5):
    """The 12 outputs of a call against the fixture, max-abs error normalised by the reference tensor's max-abs.  Where the fixture
    holds the reference's float64 run (round-6 fixtures): against FLOAT64, within max(tol, factor x the float32 reference's own distance from it)
    -- in the sharp regime the reference's float32 `weights` sit 1.1e-4 from its float64 run (neus_dtu_nonormal; SURVEY 8c measured 3.8e-5 on
    another draw), so a flat 1e-4 against the float32 vector would gate an implementation on the reference's round-off, not on its own.
    Otherwise: within tol of the float32 vector.  Returns the offending (key, error, limit) rows."""
    bad, lims = [], {}
    for k in OUTPUT_KEYS:
        if f"{tag}:out_{k}" not in fx:
            continue
        ref32 = fx[f"{tag}:out_{k}"]
        got = torch.as_tensor(out[k]).detach().cpu().reshape(ref32.shape)
        if f"{tag}:f64:out_{k}" in fx:
            ref64 = fx[f"{tag}:f64:out_{k}"]
            e, lim = relerr(got, ref64), max(tol, factor * relerr(torch.from_numpy(np.asarray(ref32)), ref64))
        else:
            e, lim = relerr(got, ref32), tol
        lims[k] = lim
        if k == "weight_max":
            # max_j w_j of a ray: |max w - max w'| <= max |w - w'| on the same scale (the largest weight), so `weights` passing its limit already
            # bounds this output by that limit; a tighter one would gate the same quantity twice with two different numbers
            lim = max(lim, lims.get("weights", 0.0))
        if not e < lim:
            bad.append((k, e, lim))
    return bad


OUTPUT_KEYS = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weights", "weight_max", "gradients",   # (weights in front of weight_max: see check_outputs)
               "gradient_error", "inside_sphere", "depth", "global_color", "delta_relight"]


FULL_TENSOR_LIMIT = 8192   # tools/gen_golden.py stores gradient tensors up to this size in full, larger ones strided
GRAD_TOL_MIN, GRAD_TOL_CAP, GRAD_SPREAD_FACTOR = 1e-4, 1e-3, 3.0
# The HIP path is held RELATIVE to the float32 reference: where the reference's own float32 evaluation is further than 1e-4 from float64,
# the product may be at most 1.5x as far (measured at C3 size: up to 1.35x on the SDF tensors, profiles/r0x_param_grad_error_table_c3.txt;
# a regression of the split-f16 arithmetic to 2x the float32 error now fails where 3x used to pass).
STRICT_SPREAD_FACTOR = 1.5


def grad_tolerance(spread, strict=False):
    """Relative tolerance (own scale: |err|_max / |tensor|_max) of one parameter-gradient tensor against the float64 reference.

    1e-4 (north_star) for every tensor, widened only where the REFERENCE's own float32 evaluation is further than that from its
    float64 evaluation of the same code at the same sample positions (``gspread`` in the fixture, measured per tensor by
    tools/gen_golden.py): there 3x the reference's own round-off (strict, the HIP path: 1.5x), never more than 1e-3 of the tensor's own
    largest entry."""
    return min(GRAD_TOL_CAP, max(GRAD_TOL_MIN, (STRICT_SPREAD_FACTOR if strict else GRAD_SPREAD_FACTOR) * float(spread)))


SCALAR_TOL_MIN = 3e-4


def scalar_tolerance(spread):
    """A ONE-entry tensor (deviation_network.variance): its gradient is a single sum in which the per-ray terms cancel to a few percent of
    their size, and what it carries is the float32 round-off of the NETWORK OUTPUTS it is made of (sdf x inv_s of several hundred), not of the
    compositor: re-forming the whole alpha / transmittance chain of the backward pass in double from the float32 sdf / normals changed the error
    on none of 22 fixture runs (round 6).  One entry is one draw of that error: over the fixtures the reference's own float32
    run sits between 8e-6 and 3.0e-4 of the float64 value and so does this implementation, the ratio of the two between 0.2 and 5.  So the
    per-fixture gate is max(3e-4, 3 x the reference's own error) under the usual cap, and the 1.5x rule is held where it means something: on
    the POPULATION of draws (check_scalar_population: RMS over all fixtures and tags)."""
    return min(GRAD_TOL_CAP, max(SCALAR_TOL_MIN, GRAD_SPREAD_FACTOR * float(spread)))


def scalar_error(fx, tag, grads, key="deviation_network.variance"):
    """(this implementation's error, the reference's own float32 error) of a one-entry gradient, both relative to the float64 value"""
    r64, r32 = float(fx[f"{tag}:g64:{key}"][0]), float(fx[f"{tag}:g:{key}"][0])
    den = max(abs(r64), 1e-300)
    return abs(float(grads[key]) - r64) / den, abs(r32 - r64) / den


def check_scalar_population(pairs, factor=STRICT_SPREAD_FACTOR):
    """The 1.5x rule for a one-entry tensor, over all fixtures: RMS of this implementation's errors <= 1.5 x RMS of the reference's own float32
    errors (pairs from scalar_error).  None when fine."""
    ours = float(np.sqrt(np.mean([a * a for a, _ in pairs])))
    ref = float(np.sqrt(np.mean([b * b for _, b in pairs])))
    return None if ours <= factor * ref else "d variance over %d runs: RMS error %.2e against %.2e for the reference's float32 run (%.2fx > %.1fx)" % (len(pairs), ours, ref, ours / ref, factor)


GRAD_OUTLIER_FRAC = 0.25    # share of a tensor's entries that may exceed the bulk tolerance (never the cap), see check_param_grads
# The HIP path (the product) is held to what profiles/r0x_param_grad_error_table.txt measures for it (strict=True below): per tensor at
# most max(STRICT_OUTLIER_MIN, 1 %) of the compared entries above the bulk tolerance -- the kink argument of check_param_grads predicts a
# handful, never a column block -- and a hard cap of 5e-4 of the tensor's own largest entry.  The loose constants above remain for the
# CPU-emulation build and the float32 torch oracle (test infrastructure whose ReLU decisions round differently on the 16-ray fixtures).
STRICT_OUTLIER_FRAC, STRICT_OUTLIER_MIN, STRICT_TOL_CAP = 0.01, 2, 5e-4


def _gate(strict):
    return (STRICT_OUTLIER_FRAC, STRICT_OUTLIER_MIN, STRICT_TOL_CAP) if strict else (GRAD_OUTLIER_FRAC, 1, GRAD_TOL_CAP)


def _allowed(n, strict):
    frac, floor, _ = _gate(strict)
    return max(floor, int(frac * n)) if n > 1 else 0


def param_grad_table(fx, tag, grads, strict=False):
    """Per-tensor comparison of full gradient tensors ``grads`` (name -> tensor) with the fixture.  Rows:
    (name, numel, err64, err32, tol, sum_err, abs_err, n_over, err_bulk) -- err64 / err32: max-abs error vs the float64 / float32
    reference entries stored in the fixture (all entries for tensors <= FULL_TENSOR_LIMIT, every grad_stride-th beyond),
    normalised by the float64 tensor's own max-abs; sum_err / abs_err: error of sum(g) and sum(|g|) over ALL entries relative to
    the reference sum(|g|); n_over: compared entries whose error exceeds tol; err_bulk: largest error once the allowed share of
    kink-affected entries (GRAD_OUTLIER_FRAC, at least one entry) is set aside."""
    s = int(fx["grad_stride"])
    names = [k[len(tag) + 5:] for k in fx if k.startswith(f"{tag}:g64:")]
    rows = []
    for k in names:
        ref64, ref32 = fx[f"{tag}:g64:{k}"], fx[f"{tag}:g:{k}"]
        full = grads[k].detach().cpu().double().reshape(-1)
        st = 1 if full.numel() <= FULL_TENSOR_LIMIT else s
        got = full[::st].numpy()
        assert got.shape == ref64.shape, (k, got.shape, ref64.shape)
        den = max(float(fx[f"{tag}:gmax64:{k}"]), 1e-300)
        e = np.abs(got - ref64) / den
        err64 = float(e.max())
        err32 = float(np.abs(got - ref32.astype(np.float64)).max()) / den
        gabs = max(float(fx[f"{tag}:gabs64:{k}"]), 1e-300)
        sum_err = abs(float(full.sum()) - float(fx[f"{tag}:gsum64:{k}"])) / gabs
        abs_err = abs(float(full.abs().sum()) - float(fx[f"{tag}:gabs64:{k}"])) / gabs
        lim = grad_tolerance(fx[f"{tag}:gspread:{k}"], strict) if e.size > 1 else scalar_tolerance(fx[f"{tag}:gspread:{k}"])
        allowed = _allowed(e.size, strict)
        bulk = float(np.sort(e)[-(allowed + 1)]) if e.size > allowed else 0.0
        rows.append((k, full.numel(), err64, err32, lim, sum_err, abs_err, int((e > lim).sum()), bulk))
    return rows


def check_param_grads(fx, tag, grads, tol=None, strict=False):
    """Gate on every parameter-gradient tensor at its OWN scale against the float64 reference:
      * HARD: every compared entry within GRAD_TOL_CAP (1e-3) of the tensor's largest entry;
      * BULK: at least 1 - GRAD_OUTLIER_FRAC of the entries within grad_tolerance (1e-4, or 3x the reference's own float32 round-off on
        that tensor);
      * sum(g) and sum(|g|) over ALL entries (catches a wrong entry the stride skipped) within 3x the tensor tolerance of sum(|g|).
    Why a bulk rule and not 1e-4 on every entry: the colour / relight stacks are ReLU networks.  The DTU-size fixtures hold 4.7 M
    ReLU decisions on 2048 sample points, and a handful of pre-activations sit within float32 round-off of the kink (dtu_sharp/jit:
    relight rl_mlp.1 unit 247 at point 814 is -1.1e-7 in the reference's float64 run, -3.2e-8 in its float32 run).  Whichever side
    an implementation rounds to, that single sample's whole contribution moves: one entry of the layer's own bias / weight gradient
    by up to 6e-4 of the tensor max on these 16-ray fixtures, and a fraction of the entries of the layers below it by 1e-4..3e-4.
    No float32 implementation is exempt (the derivative is discontinuous there); the cap bounds the effect.
    Returns the offending rows.  ``tol`` (optional) raises the floor of the bulk tolerance.  ``strict``: the HIP gate (constants above)."""
    bad = []
    cap = _gate(strict)[2]
    for k, n, err64, err32, lim, sum_err, abs_err, n_over, bulk in param_grad_table(fx, tag, grads, strict):
        if tol is not None:
            lim = min(cap, max(lim, tol))
        lim = min(lim, cap)
        slim = min(GRAD_TOL_CAP, 3.0 * lim)   # sums: same-sign round-off adds up over the entries while sum(|g|) can be far below n * max
        if not (err64 <= cap and bulk <= lim and sum_err <= slim and abs_err <= slim):
            bad.append((k, err64, bulk, lim, sum_err, abs_err))
    return bad


def check_grads_full(ref64, ref32, got, strict=True, rel_max=None):
    """EVERY entry of every gradient tensor (no stride) against a float64 evaluation ``ref64`` (name -> tensor) of the same algorithm
    at the same inputs -- the oracle run live by the GPU tests; ``ref32`` is its float32 evaluation, whose distance from float64
    calibrates the per-tensor tolerance exactly like the fixtures' ``gspread``.  Same rule as check_param_grads.  Returns offenders.
    ``rel_max`` (full-size batches): in addition the LARGEST error of every tensor of the smooth part of the model -- the softplus SDF
    network, the variance, d rays -- must be within max(1e-4, rel_max x the float32 oracle's largest error on that tensor).  The ReLU stacks
    keep the bulk rule + cap: even at 4096 rays ONE flipped unit at one sample shows in its row (measured: relight rl_mlp.2, 3 entries at
    1.3e-4 where every other entry of the tensor is within 1e-6; profiles/r05_param_grad_error_table_c4.txt)."""
    bad = []
    cap = _gate(strict)[2]
    for k, r64 in ref64.items():
        r64 = r64.detach().double().reshape(-1)
        den = max(float(r64.abs().max()), 1e-300)
        e = (got[k].detach().cpu().double().reshape(-1) - r64).abs() / den
        spread = float((ref32[k].detach().double().reshape(-1) - r64).abs().max()) / den
        lim = min(cap, grad_tolerance(spread, strict) if e.numel() > 1 else scalar_tolerance(spread))
        allowed = _allowed(e.numel(), strict)
        bulk = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        ok = float(e.max()) <= cap and bulk <= lim
        if rel_max is not None and k.split(".")[0] in ("sdf_network", "deviation_network", "rays_o", "rays_d"):
            ok = ok and float(e.max()) <= max(GRAD_TOL_MIN, rel_max * spread)
        if not ok:
            bad.append((k, float(e.max()), bulk, lim, int((e > lim).sum()), spread))
    return bad


def check_input_grad(fx, tag, key, got, strict=False):
    """d rays_o / d rays_d / d near / d far against the float64 reference at own scale under the same rule as check_param_grads
    (hard cap on every entry, bulk tolerance from the reference's own float32 spread measured from the two stored runs).
    Returns None when within the gate, else (key, err_max, err_bulk, tol)."""
    ref64, ref32 = fx[f"{tag}:f64:{key}"], fx[f"{tag}:{key}"]
    den = max(float(np.abs(ref64).max()), 1e-300)
    spread = float(np.abs(ref32.astype(np.float64) - ref64).max()) / den
    e = np.abs(torch.as_tensor(got).detach().cpu().double().numpy().reshape(ref64.shape) - ref64).reshape(-1) / den
    cap = _gate(strict)[2]
    lim = min(cap, grad_tolerance(spread, strict))
    allowed = max(1, _allowed(e.size, strict))
    bulk = float(np.sort(e)[-(allowed + 1)])
    if float(e.max()) <= cap and bulk <= lim:
        return None
    return key, float(e.max()), bulk, lim


def format_grad_table(title, rows):
    lines = [f"# {title}", "%-42s %8s %10s %10s %10s %9s %6s %10s %10s" % ("tensor", "numel", "err_vs_f64", "err_bulk", "err_vs_f32", "tol", "n>tol", "sum_err", "abs_err")]
    for k, n, e64, e32, lim, se, ae, nover, bulk in rows:
        lines.append("%-42s %8d %10.2e %10.2e %10.2e %9.1e %6d %10.2e %10.2e" % (k, n, e64, bulk, e32, lim, nover, se, ae))
    lines.append("worst err_vs_f64 %.2e (cap %.0e), worst err_bulk/tol %.2f" % (max(r[2] for r in rows), GRAD_TOL_CAP, max(r[8] / r[4] for r in rows)))
    return "\n".join(lines)
