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


def _run(library, device, name="tiny_sharp", cos_anneal=0.3, bg=(0.2, 0.5, 0.7)):
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    g = torch.Generator().manual_seed(123)
    z = torch.from_numpy(fx["jit:z_vals"])
    o, d = torch.from_numpy(fx["rays_o"]), torch.from_numpy(fx["rays_d"])
    near, far = torch.from_numpy(fx["jit:near"]), torch.from_numpy(fx["jit:far"])
    bgt = torch.tensor(bg)
    # oracle (float64 for a clean reference)
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    o64, d64 = o.double().requires_grad_(True), d.double().requires_grad_(True)
    out_o = O.render(P64, ocfg, o64, d64, near.double(), far.double(), z_vals=z.double(), cos_anneal_ratio=cos_anneal,
                     background_rgb=bgt.double())
    coefs = {k: torch.randn(out_o[k].shape, generator=g, dtype=torch.float64) for k in KEYS if k in out_o}
    coefs["weights"] *= 3.0
    coefs["gradient_error"] = coefs["gradient_error"] * 5.0
    L_o = sum((out_o[k] * coefs[k]).sum() for k in coefs)
    L_o.backward()
    # native
    r = N.make_renderer(ocfg, P, library, device)
    on, dn = o.to(device).requires_grad_(True), d.to(device).requires_grad_(True)
    out_n = r(on, dn, near.to(device), far.to(device), z_vals=z.to(device), cos_anneal_ratio=cos_anneal, background_rgb=bgt)
    L_n = sum((out_n[k] * coefs[k].float().to(device).reshape(out_n[k].shape)).sum() for k in coefs)
    L_n.backward()
    errs = {}
    for k in coefs:
        errs["out:" + k] = G.relerr(out_n[k].detach().cpu().reshape(out_o[k].shape), out_o[k].detach())
    for k, p in r.named_parameters():
        ref = P64[k].grad
        e = ((p.grad.detach().cpu().double() - ref).abs() / float(ref.abs().max())).reshape(-1)   # own scale per tensor
        allowed = max(1, int(G.GRAD_OUTLIER_FRAC * e.numel())) if e.numel() > 1 else 0
        errs["grad:" + k] = float(e.max())
        errs["bulk:" + k] = float(torch.sort(e).values[-(allowed + 1)]) if e.numel() > allowed else 0.0
    errs["grad:rays_o"] = G.relerr(on.grad.cpu(), o64.grad)
    errs["grad:rays_d"] = G.relerr(dn.grad.cpu(), d64.grad)
    return errs


def _check(errs):
    # Parameter gradients at own scale per tensor.  Every per-sample output carries an O(1) random cotangent here, so a single
    # sample is a far larger share of a gradient entry than under the training loss, and the ReLU-kink events described in
    # tests/_golden.py (check_param_grads) move individual entries by up to ~6e-3 of the tensor max on the 16-ray DTU fixture
    # (identically on the CPU-emulation and the HIP build).  A wrong backward formula moves whole tensors by O(1): the bulk rule
    # (all but GRAD_OUTLIER_FRAC of the entries within 5e-4: the layers below a flipped unit all move a little) is the gate, the per-entry cap only bounds the kink effect.
    loose = {"out:weights": 5e-4, "out:weight_max": 5e-4, "out:cdf_fine": 5e-4}   # see test_hip_parity.test_against_oracle_larger_batch
    lim = lambda k: loose.get(k, 2e-2 if k.startswith("grad:color") or k.startswith("grad:relight") or k.startswith("grad:sdf") or k.startswith("grad:dev")
                              else (5e-4 if k.startswith("bulk:") else (2e-4 if k.startswith("grad:") else 1e-4)))
    bad = {k: e for k, e in errs.items() if not e < lim(k)}
    assert not bad, bad


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp"])
def test_all_outputs_backward_emu(name):
    _check(_run(N.EMU_LIB, "cpu", name))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp", "dtu_sharp"])
def test_all_outputs_backward_hip(name):
    _check(_run(None, "cuda:0", name))
