"""Shared driver for the native renderer tests (CPU emulation build here, HIP build on the GPU box)."""
import os

import torch

import _golden as G
import color_neus_amd as cn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# CNR_EMU_LIB: an alternative build of the CPU emulation (tools/run_sanitizers.sh points it at the ASan / UBSan build)
EMU_LIB = os.environ.get("CNR_EMU_LIB") or os.path.join(ROOT, "tests", "_build", "libcolorneus_emu.so")


def render_config_from_oracle(ocfg) -> cn.RenderConfig:
    s, c, r = ocfg.sdf, ocfg.color, ocfg.relight
    kw = dict(type=ocfg.type, n_samples=ocfg.n_samples, n_importance=ocfg.n_importance, n_outside=ocfg.n_outside, up_sample_steps=ocfg.up_sample_steps,
              perturb=ocfg.perturb, sdf_d_out=s.d_out, sdf_d_hidden=s.d_hidden, sdf_n_layers=s.n_layers,
              sdf_skip_in=list(s.skip_in), sdf_multires=s.multires, sdf_bias=s.bias, sdf_scale=s.scale,
              sdf_weight_norm=s.weight_norm, col_d_feature=c.d_feature, col_mode=c.mode, col_d_in=c.d_in,
              col_d_hidden=c.d_hidden, col_n_layers=c.n_layers, col_weight_norm=c.weight_norm,
              col_multires_view=c.multires_view, col_squeeze_out=c.squeeze_out, init_val=ocfg.init_val)
    if r is not None:
        kw.update(rel_d_in=r.d_in, rel_d_hidden=r.d_hidden, rel_n_layers=r.n_layers, rel_y_in_layer=r.y_in_layer,
                  rel_multires_view=r.multires_view, rel_include_grad=r.include_grad, rel_inv_sigmoid=r.inv_sigmoid)
    return cn.RenderConfig(**kw)


def make_renderer(ocfg, P, library, device):
    rc = render_config_from_oracle(ocfg)
    cls = cn.ColorNeuSRenderer if ocfg.type == "Color_NeuS" else cn.NeuSRenderer
    r = cls(rc, library=library)
    missing, unexpected = r.load_state_dict({k: v.float() for k, v in P.items()}, strict=True)
    return r.to(device)


def run_native(name, tag, library, device, fixed_z=True, rays_grad=True, nearfar_grad=False):
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    r = make_renderer(ocfg, P, library, device)
    o = torch.from_numpy(fx["rays_o"]).to(device).requires_grad_(rays_grad)
    d = torch.from_numpy(fx["rays_d"]).to(device).requires_grad_(rays_grad)
    near, far = torch.from_numpy(fx[f"{tag}:near"]).to(device), torch.from_numpy(fx[f"{tag}:far"]).to(device)
    if nearfar_grad:
        near.requires_grad_(True)
        far.requires_grad_(True)
    z = torch.from_numpy(fx[f"{tag}:z_vals"]).to(device) if fixed_z else None
    ckw = G.call_kwargs(fx, device)    # cos_anneal_ratio / background_rgb of the *_anneal fixtures
    if ocfg.n_outside > 0:
        # N_OUTSIDE > 0: the background samples take a second draw from the CPU generator (NeuS.py:335) -- replay the fixture's seed
        # (tools/gen_golden.py: jitter seed 2) instead of patching torch.rand
        if tag == "jit":
            torch.manual_seed(2)
            out = r(o, d, near, far, z_vals=z)
        else:
            out = r(o, d, near, far, z_vals=z, perturb_overwrite=0)
    elif fixed_z:
        out = r(o, d, near, far, z_vals=z, **ckw)
    elif f"{tag}:t_rand" in fx:
        # feed the fixture's jitter draw through the same CPU-generator call the module makes
        t = torch.from_numpy(fx[f"{tag}:t_rand"])
        orig = torch.rand
        try:
            torch.rand = lambda *a, **k: t.clone()
            out = r(o, d, near, far, **ckw)
        finally:
            torch.rand = orig
    else:
        out = r(o, d, near, far, perturb_overwrite=0, **ckw)
    loss, _ = cn.compute_loss(out, torch.from_numpy(fx["rgb_gt"]).to(device), torch.from_numpy(fx["mask"]).to(device))
    loss.backward()
    grads = {k: p.grad for k, p in r.named_parameters()}
    if nearfar_grad:
        return fx, r, out, loss, grads, o, d, near, far
    return fx, r, out, loss, grads, o, d

