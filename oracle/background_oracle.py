"""CPU ORACLE for the N_OUTSIDE > 0 path (NeRF++ background of NeuS).  TEST INFRASTRUCTURE ONLY.

Plain-torch restatement of lib/models/renderers/NeuS.py:95-134 (render_core_outside), :313-369 (background samples, z_vals_feed),
:236-292 / Color_NeuS.py:66-138 (render_core with background_alpha / background_sampled_color) and of the NeRF network
(lib/models/renderers/fields.py:192-274).  Pinned by the fixtures tiny_outside / tiny_neus_outside captured from the imported reference
(tools/gen_golden.py): tests/test_oracle_golden.py holds this file to them.  Only tests/ may import it; the product path
(color-neus_amd/background.py) runs the same arithmetic in the HIP library and fails loudly without it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _embed(x, multires):
    """get_embedder(multires, input_dims=d): [x, sin(2^k x), cos(2^k x)]_k (PositionEncoding.py:51-76)."""
    out = [x]
    for k in range(multires):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


class NeRF(nn.Module):
    """Parameters and forward of the reference's NeRF background network built with its defaults (NeuS.__init__ falls back to NeRF()
    for every cfg, NeuS.py:87-91): 8 x 256 ReLU layers on PE-10 of a 4-vector (x / r, 1 / r), skip at 4, view branch on PE-4."""

    def __init__(self, D=8, W=256, d_in=4, d_in_view=3, multires=10, multires_view=4, skips=(4,)):
        super().__init__()
        self.multires, self.multires_view, self.skips = multires, multires_view, tuple(skips)
        ch = d_in * (1 + 2 * multires)
        ch_view = d_in_view * (1 + 2 * multires_view)
        self.pts_linears = nn.ModuleList([nn.Linear(ch, W)] + [nn.Linear(W + ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(ch_view + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)

    def _lin(self, lin, x, relu):
        y = lin(x)
        return F.relu(y) if relu else y

    def forward(self, pts, views):
        e = _embed(pts, self.multires)
        h = e
        for i, lin in enumerate(self.pts_linears):
            h = self._lin(lin, h, True)
            if i in self.skips:
                h = torch.cat([e, h], -1)
        density = self._lin(self.alpha_linear, h, False)
        h = torch.cat([self._lin(self.feature_linear, h, False), _embed(views, self.multires_view)], -1)
        for lin in self.views_linears:
            h = self._lin(lin, h, True)
        return density, self._lin(self.rgb_linear, h, False)


def outside_samples(far, n_outside, n_samples, perturb):
    """z of the background samples: inverse-depth spacing beyond ``far`` (NeuS.py:315-338); draws torch.rand([R, n_outside]) when perturb."""
    dev = far.device
    z = torch.linspace(1e-3, 1.0 - 1.0 / (n_outside + 1.0), n_outside).to(dev)
    if perturb > 0:
        mids = 0.5 * (z[1:] + z[:-1])
        upper, lower = torch.cat([mids, z[-1:]], -1), torch.cat([z[:1], mids], -1)
        z = lower[None, :] + (upper - lower)[None, :] * torch.rand([far.shape[0], n_outside]).to(dev)
    return far.reshape(-1, 1) / torch.flip(z, dims=[-1]) + 1.0 / n_samples


def render_outside(nerf, rays_o, rays_d, z_feed, sample_dist):
    """render_core_outside (NeuS.py:95-134) on the merged sample positions: per-sample alpha and colour of the background."""
    dists = torch.cat([z_feed[:, 1:] - z_feed[:, :-1], torch.full_like(z_feed[:, :1], sample_dist)], -1)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * (z_feed + dists * 0.5)[..., None]
    r = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts4 = torch.cat([pts / r, 1.0 / r], dim=-1)
    n, m = z_feed.shape
    density, rgb = nerf(pts4.reshape(-1, 4), rays_d[:, None, :].expand(n, m, 3).reshape(-1, 3))
    alpha = 1.0 - torch.exp(-F.softplus(density.reshape(n, m)) * dists)
    return alpha, torch.sigmoid(rgb).reshape(n, m, 3)


def _exclusive_transmittance(alpha):
    ones = torch.ones_like(alpha[:, :1])
    return torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]


def composite_with_background(type_, rays_o, rays_d, z, sample_dist, inv_s, sdf, gradients, color, gcolor, delta_relight, bg_alpha, bg_color,
                              z_feed, cos_anneal_ratio, background_rgb):
    """The tail of render_core when a background is present (NeuS.py:236-292, Color_NeuS.py:66-138): S-density alpha from the
    library's per-sample sdf / normals, inside / outside mixing, compositing over M + N_OUTSIDE samples.  Returns the reference dict."""
    n, M = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], sample_dist)], -1)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * (z + dists * 0.5)[..., None]
    true_cos = (rays_d[:, None, :] * gradients).sum(-1)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    prev_cdf = torch.sigmoid((sdf - iter_cos * dists * 0.5) * inv_s)
    next_cdf = torch.sigmoid((sdf + iter_cos * dists * 0.5) * inv_s)
    alpha_in = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
    pn = torch.linalg.norm(pts, ord=2, dim=-1)
    inside = (pn < 1.0).float().detach()
    relax = (pn < 1.2).float().detach()
    alpha = torch.cat([alpha_in * inside + bg_alpha[:, :M] * (1.0 - inside), bg_alpha[:, M:]], -1)
    mixed = torch.cat([color * inside[..., None] + bg_color[:, :M] * (1.0 - inside)[..., None], bg_color[:, M:]], 1)
    w = alpha * _exclusive_transmittance(alpha)
    wsum = w.sum(-1, keepdim=True)
    col = (mixed * w[..., None]).sum(1)
    if background_rgb is not None:
        col = col + background_rgb * (1.0 - wsum)
    eik_num, eik_den = (relax * (torch.linalg.norm(gradients, ord=2, dim=-1) - 1.0) ** 2).sum(), relax.sum()
    gerr = eik_num / (eik_den + 1e-5)
    out = {"color_fine": col, "s_val": (1.0 / inv_s).expand(n, M).mean(-1, keepdim=True), "cdf_fine": prev_cdf, "weight_sum": wsum,
           "weight_max": torch.max(w, dim=-1, keepdim=True)[0], "gradients": gradients, "weights": w, "gradient_error": gerr,
           "inside_sphere": inside, "depth": torch.sum(w * z_feed, -1),
           "eik_sums": torch.stack([eik_num, eik_den])}   # {sum relax*(|g|-1)^2, sum relax}: what ray-sharded training all-reduces
    if type_ == "Color_NeuS":
        w_in = alpha_in * _exclusive_transmittance(alpha_in)          # global colour is composited with the foreground weights only
        out["global_color"] = (gcolor * w_in[..., None]).sum(1)
        out["delta_relight"] = delta_relight
    return out


def nerf_from_params(P, dtype=torch.float32):
    """The NeRF module filled from the ``nerf.*`` entries of a flat state dict."""
    nerf = NeRF()
    nerf.load_state_dict({k[len("nerf."):]: v for k, v in P.items() if k.startswith("nerf.")}, strict=True)
    return nerf.to(dtype)


def render(P, cfg, rays_o, rays_d, near, far, z_vals, t_out=None, cos_anneal_ratio=0.0, background_rgb=None):
    """NeuS.forward with N_OUTSIDE > 0 at GIVEN foreground sample positions z_vals (NeuS.py:313-408): background samples from ``far`` and the
    uniform draw ``t_out`` [R, n_outside] (None: no perturbation), background network on the merged positions, render_core with mixing.
    ``P``: flat state dict incl. ``nerf.*``; differentiable w.r.t. every entry, rays_o / rays_d and far."""
    from . import colorneus_oracle as O
    dt = z_vals.dtype
    R, M = z_vals.shape
    n_out = cfg.n_outside
    sample_dist = 2.0 / cfg.n_samples
    zz = torch.linspace(1e-3, 1.0 - 1.0 / (n_out + 1.0), n_out).to(dt)
    if t_out is not None:
        mids = 0.5 * (zz[1:] + zz[:-1])
        upper, lower = torch.cat([mids, zz[-1:]], -1), torch.cat([zz[:1], mids], -1)
        zz = lower[None, :] + (upper - lower)[None, :] * t_out.to(dt)
    z_out = far.reshape(-1, 1) / torch.flip(zz, dims=[-1]) + 1.0 / cfg.n_samples
    z_feed, _ = torch.sort(torch.cat([z_vals, z_out.expand(R, n_out)], dim=-1), dim=-1)
    nerf = nerf_from_params(P, dt)
    # functional view of the module's parameters so that autograd reaches the entries of P
    bg_alpha, bg_color = torch.func.functional_call(_Outside(nerf), {"nerf." + k: P["nerf." + k] for k, _ in nerf.named_parameters()},
                                                     (rays_o, rays_d, z_feed, sample_dist))
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * (z_vals + dists * 0.5)[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(R, M, 3).reshape(-1, 3)
    sdf, feat, g = O.sdf_forward(P, cfg.sdf, pts, want_grad=True)
    gcol = drgb = None
    if cfg.type == "Color_NeuS":
        gcol = O.color_forward(P, cfg.color, pts, g, dirs, feat)
        relit, drgb = O.relight_forward(P, cfg.relight, gcol, pts, dirs, g)
        sampled = relit
    else:
        sampled = O.color_forward(P, cfg.color, pts, g, dirs, feat)
    inv_s = torch.exp(P["deviation_network.variance"] * 10.0).clamp(1e-6, 1e6)
    out = composite_with_background(cfg.type, rays_o, rays_d, z_vals, sample_dist, inv_s, sdf.reshape(R, M), g.reshape(R, M, 3), sampled.reshape(R, M, 3),
                                    gcol.reshape(R, M, 3) if gcol is not None else None, drgb.reshape(R, M, 3) if drgb is not None else None,
                                    bg_alpha, bg_color, z_feed, cos_anneal_ratio, background_rgb)
    out["z_vals"] = z_vals
    return out


class _Outside(nn.Module):
    def __init__(self, nerf):
        super().__init__()
        self.nerf = nerf

    def forward(self, rays_o, rays_d, z_feed, sample_dist):
        return render_outside(self.nerf, rays_o, rays_d, z_feed, sample_dist)
