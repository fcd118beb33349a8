"""N_OUTSIDE > 0: the NeRF++ background of NeuS (lib/models/renderers/NeuS.py:95-134, 313-369, fields.py:192-274).

No shipped configuration enables it (N_OUTSIDE is absent from every config/*.yml, NeuS.py:82).  The foreground -- sampler, SDF / colour /
relight stacks and their backward -- runs in the HIP library: the render call hands out its per-sample outputs (sdf, normals, colours:
cnr_render_outputs.*_samples) and takes their gradients back (cnr_render_out_grads.*_samples).  The background NETWORK (the 8 x 256 ReLU
stack, its skip layer, the view branch and the two heads: 12 nn.Linear layers) runs on the library's layer / weight-gradient kernels through
cnr_linear_forward / cnr_linear_backward (HipLinear below) whenever a render library is attached (NeRF.library); what stays in torch is the
glue around it: the positional encodings, the concatenations, and the [R, M + N_OUTSIDE] inside / outside alpha mixing."""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F


class HipLinear(torch.autograd.Function):
    """y = act(x W^T + b) through cnr_linear_forward / cnr_linear_backward (the render library's layer GEMM + weight-gradient GEMM)."""

    @staticmethod
    def forward(ctx, lib, x, weight, bias, relu):
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous().float()
        w, b = weight.detach().contiguous().float(), (bias.detach().contiguous().float() if bias is not None else None)
        n, k, n_out = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty(n, n_out, dtype=torch.float32, device=x2.device)
        if n > 0:
            nb = lib.lib.cnr_linear_scratch_bytes(n, k, n_out, 0)
            scratch = torch.empty(nb, dtype=torch.uint8, device=x2.device)
            stream = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream) if x2.is_cuda else C.c_void_p(0)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            lib.check(lib.lib.cnr_linear_forward(p(x2), n, k, p(w), p(b), n_out, int(relu), p(y), p(scratch), nb, stream), "cnr_linear_forward")
        ctx.lib, ctx.relu, ctx.has_bias, ctx.xshape = lib, bool(relu), bias is not None, x.shape
        ctx.save_for_backward(x2, w, y)
        return y.reshape(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        lib = ctx.lib
        n, k, n_out = x2.shape[0], x2.shape[1], w.shape[0]
        dy2 = dy.reshape(-1, n_out).contiguous().float()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        dW = torch.empty_like(w)
        db = torch.empty(n_out, dtype=torch.float32, device=w.device) if ctx.has_bias else None
        if n > 0:
            nb = lib.lib.cnr_linear_scratch_bytes(n, k, n_out, 1)
            scratch = torch.empty(nb, dtype=torch.uint8, device=x2.device)
            stream = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream) if x2.is_cuda else C.c_void_p(0)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            lib.check(lib.lib.cnr_linear_backward(p(x2), p(y), p(dy2), n, k, p(w), n_out, int(ctx.relu), p(dx), p(dW), p(db), p(scratch), nb, stream),
                      "cnr_linear_backward")
        else:
            dW.zero_()
            if db is not None:
                db.zero_()
        return None, (dx.reshape(ctx.xshape) if dx is not None else None), dW, db, None


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
        self.library = None   # a _lib.RenderLibrary: the layers then run on its kernels (set by the renderer that owns this network)

    def _lin(self, lin, x, relu):
        if self.library is not None:
            return HipLinear.apply(self.library, x, lin.weight, lin.bias, relu)
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
