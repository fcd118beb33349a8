"""Backward for EVERY differentiable output (not only the four the trainer's loss consumes), background_rgb and a non-zero
cos_anneal_ratio: native path vs autograd of the oracle on a random linear functional of all outputs.
CPU: emulation build; GPU (-m gpu): HIP build."""
import os

import pytest
import torch

import _golden as G
import _native as N
from oracle import colorneus_oracle as O

KEYS = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights", "gradient_error", "depth",
        "global_color", "delta_relight"]


def _run(library, device, name="tiny_sharp", cos_anneal=0.3, bg=(0.2, 0.5, 0.7), strict=False):
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    g = torch.Generator().manual_seed(123)
    z = torch.from_numpy(fx["jit:z_vals"])
    o, d = torch.from_numpy(fx["rays_o"]), torch.from_numpy(fx["rays_d"])
    near, far = torch.from_numpy(fx["jit:near"]), torch.from_numpy(fx["jit:far"])
    bgt = torch.tensor(bg)
    # oracle in float64 (the reference value) and in float32 (= the reference's own arithmetic: its distance from float64 calibrates the
    # tolerances per tensor, like ``gspread`` in the fixtures)
    ref, coefs = {}, None
    for dt in (torch.float64, torch.float32):
        Pd = {k: v.to(dt).clone().requires_grad_(True) for k, v in P.items()}
        od, dd = o.to(dt).clone().requires_grad_(True), d.to(dt).clone().requires_grad_(True)
        out_o = O.render(Pd, ocfg, od, dd, near.to(dt), far.to(dt), z_vals=z.to(dt), cos_anneal_ratio=cos_anneal, background_rgb=bgt.to(dt))
        if coefs is None:
            coefs = {k: torch.randn(out_o[k].shape, generator=g, dtype=torch.float64) for k in KEYS if k in out_o}
            coefs["weights"] *= 3.0
            coefs["gradient_error"] = coefs["gradient_error"] * 5.0
        L_o = sum((out_o[k] * coefs[k].to(dt)).sum() for k in coefs)
        L_o.backward()
        gr = {k: v.grad.detach() for k, v in Pd.items()}
        gr["rays_o"], gr["rays_d"] = od.grad.detach(), dd.grad.detach()
        ref[dt] = ({k: out_o[k].detach() for k in coefs}, gr)
    out64, g64 = ref[torch.float64]
    out32, g32 = ref[torch.float32]
    # native
    r = N.make_renderer(ocfg, P, library, device)
    on, dn = o.to(device).requires_grad_(True), d.to(device).requires_grad_(True)
    out_n = r(on, dn, near.to(device), far.to(device), z_vals=z.to(device), cos_anneal_ratio=cos_anneal, background_rgb=bgt)
    L_n = sum((out_n[k] * coefs[k].float().to(device).reshape(out_n[k].shape)).sum() for k in coefs)
    L_n.backward()
    got = {(k[len("renderer."):] if k.startswith("renderer.") else k): p.grad for k, p in r.named_parameters()}
    got["rays_o"], got["rays_d"] = on.grad, dn.grad
    bad = {}
    for k in coefs:
        e = G.relerr(out_n[k].detach().cpu().reshape(out64[k].shape), out64[k])
        e32 = G.relerr(out32[k].double(), out64[k])
        # per-sample weights / cdf at inv_s = 665 are the rounding-sensitive outputs: 3 x the float32 oracle's own distance from float64 where
        # that exceeds 1e-4, never more than 5e-4 (strict) -- the emulation build keeps the round-2 constant
        lim = min(5e-4, max(1e-4, 3.0 * e32)) if strict else (5e-4 if k in ("weights", "weight_max", "cdf_fine") else 1e-4)
        if not e < lim:
            bad["out:" + k] = (e, lim)
    # Parameter gradients and d rays at own scale per tensor.  Every per-sample output carries an O(1) random cotangent here, so a single
    # sample is a far larger share of a gradient entry than under the training loss: a ReLU unit whose pre-activation sits within float32
    # round-off of zero (tests/_golden.py, check_param_grads) moves the entries of ITS row by up to 6.1e-3 of the tensor max on the 16-ray DTU
    # fixture (colour lin0: one bias entry, the 262 weight entries of that unit; measured identically on the HIP and the emulation build,
    # tools/gate_probe.py).  So the hard cap only bounds the kink effect (KINK_CAP), and the gate is the BULK rule: on the HIP build all but
    # max(2, 1 %) of a tensor's entries within max(1e-4, 3 x the float32 oracle's own error) -- a wrong backward formula moves whole tensors
    # by O(1).  The emulation build keeps the looser round-2 constants (25 %, 5e-4).
    KINK_CAP = 1e-2
    for k, r64 in g64.items():
        r64 = r64.double().reshape(-1)
        den = max(float(r64.abs().max()), 1e-300)
        e = (got[k].detach().cpu().double().reshape(-1) - r64).abs() / den
        spread = float((g32[k].double().reshape(-1) - r64).abs().max()) / den
        allowed = G._allowed(e.numel(), strict)
        bulk = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
        lim = min(G.STRICT_TOL_CAP, G.grad_tolerance(spread, strict)) if strict else 5e-4
        cap = KINK_CAP if (strict or k.split(".")[0] in ("color_network", "relight_network", "sdf_network", "deviation_network")) else 2e-4
        if not (float(e.max()) <= cap and bulk <= lim):
            bad["grad:" + k] = (float(e.max()), bulk, lim)
    return bad


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp"])
def test_all_outputs_backward_emu(name):
    bad = _run(N.EMU_LIB, "cpu", name)
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp", "dtu_sharp"])
def test_all_outputs_backward_hip(name):
    bad = _run(None, "cuda:0", name, strict=True)   # the product path: strict gate (1 % outliers, tolerance from the float32 oracle's spread)
    assert not bad, bad
