"""CPU ORACLE for the Color-NeuS volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch functional restatement (plain torch tensor ops, any
dtype, CPU or GPU tensors) of the algorithm the reference implements in

    lib/models/renderers/NeuS.py        (sampler, up-sampling, compositing)
    lib/models/renderers/Color_NeuS.py  (render_core with global colour + relight)
    lib/models/renderers/fields.py      (SDF / colour / relight / variance MLPs)
    lib/models/tools/ray_utils.py       (sample_pdf, near_far_from_sphere)
    lib/models/tools/PositionEncoding.py, lib/utils/transform.py (inverse_sigmoid)
    lib/models/NeuS_Trainer.py:129-171  (compute_loss, the consumer of the path)

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
against golden vectors captured from the imported reference (tools/gen_golden.py,
fixtures under tests/golden/), and tests/test_reference_dropin.py re-checks the drop-in
live against /root/reference when that checkout is present.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (color-neus_amd/) never does: it fails loudly when
the HIP library is missing.

Parameters are passed as a flat dict keyed by the reference's state_dict names
(``sdf_network.lin0.weight_g`` ...), so a reference checkpoint is directly usable.
Differences in *structure* (not semantics) from the reference:
  * the SDF input-gradient is the explicit analytic chain (reverse sweep through
    the layers) instead of a second forward + torch.autograd.grad; it is built of
    differentiable torch ops, so autograd through it yields the same second-order
    terms (checked to ~1e-15 in float64 against the reference);
  * the SDF forward is evaluated once per fine point, not twice.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration (mirrors the MODEL.RENDERER sub-tree of config/*.yml)
# --------------------------------------------------------------------------------------
@dataclass
class SDFConfig:  # fields.py:15-29
    d_in: int = 3
    d_out: int = 257
    d_hidden: int = 256
    n_layers: int = 8
    skip_in: List[int] = field(default_factory=lambda: [4])
    multires: int = 6
    bias: float = 0.5
    scale: float = 3.0
    geometric_init: bool = True
    weight_norm: bool = True
    inside_outside: bool = False


@dataclass
class ColorConfig:  # fields.py:126-134
    d_feature: int = 256
    mode: str = "idr"
    d_in: int = 9
    d_out: int = 3
    d_hidden: int = 256
    n_layers: int = 4
    weight_norm: bool = True
    multires_view: int = 4
    squeeze_out: bool = True


@dataclass
class RelightConfig:  # fields.py:296-303
    d_in: int = 6
    d_out: int = 3
    d_hidden: int = 256
    n_layers: int = 4
    y_in_layer: int = 3
    multires_view: int = 4
    include_grad: bool = True
    inv_sigmoid: bool = True


@dataclass
class RenderConfig:  # NeuS.py:80-85
    type: str = "Color_NeuS"
    n_samples: int = 64
    n_importance: int = 64
    n_outside: int = 0
    up_sample_steps: int = 4
    perturb: float = 1.0
    N: int = 64
    sdf: SDFConfig = field(default_factory=SDFConfig)
    color: ColorConfig = field(default_factory=ColorConfig)
    relight: Optional[RelightConfig] = field(default_factory=RelightConfig)
    init_val: float = 0.3


def _get(node, key, default):
    if node is None:
        return default
    if hasattr(node, "get"):
        return node.get(key, default)
    return getattr(node, key, default)


def config_from_node(node) -> RenderConfig:
    """Build a RenderConfig from a yacs CfgNode / dict with the reference's upper-case keys."""
    s, c, r, d = (_get(node, k, None) for k in ("SDF", "COLOR", "RELIGHT", "DEVIATION"))
    typ = _get(node, "TYPE", "Color_NeuS")
    sdf = SDFConfig(_get(s, "D_IN", 3), _get(s, "D_OUT", 257), _get(s, "D_HIDDEN", 256), _get(s, "N_LAYERS", 8),
                    list(_get(s, "SKIP_IN", [4])), _get(s, "MULTIRES", 6), _get(s, "BIAS", 0.5),
                    _get(s, "SCALE", 3.0), _get(s, "GEOMETRIC_INIT", True), _get(s, "WEIGHT_NORM", True),
                    _get(s, "INSIDE_OUTSIDE", False))
    col = ColorConfig(_get(c, "D_FEATURE", 256), _get(c, "MODE", "idr"), _get(c, "D_IN", 9), _get(c, "D_OUT", 3),
                      _get(c, "D_HIDDEN", 256), _get(c, "N_LAYERS", 4), _get(c, "WEIGHT_NORM", True),
                      _get(c, "MULTIRES_VIEW", 4), _get(c, "SQUEEZE_OUT", True))
    rel = None
    if typ == "Color_NeuS":
        rel = RelightConfig(_get(r, "D_IN", 6), _get(r, "D_OUT", 3), _get(r, "D_HIDDEN", 256),
                            _get(r, "N_LAYERS", 4), _get(r, "Y_IN_LAYER", 3), _get(r, "MULTIRES_VIEW", 4),
                            _get(r, "INCLUDE_GRAD", True), _get(r, "INV_SIGMOID", True))
    return RenderConfig(typ, _get(node, "N_SAMPLES", 64), _get(node, "N_IMPORTANCE", 64),
                        _get(node, "N_OUTSIDE", 0), _get(node, "UP_SAMPLE_STEPS", 4), _get(node, "PERTURB", 1.0),
                        _get(node, "N", 64), sdf, col, rel, _get(d, "INIT_VAL", 0.3))


def embed_dim(multires: int, d: int = 3) -> int:
    return d * (1 + 2 * multires) if multires > 0 else d


def sdf_layer_dims(cfg: SDFConfig):
    """[(in, out)] per linear layer; fields.py:31-50."""
    d0 = embed_dim(cfg.multires, cfg.d_in)
    dims = [d0] + [cfg.d_hidden] * cfg.n_layers + [cfg.d_out]
    out = []
    for l in range(len(dims) - 1):
        o = dims[l + 1] - d0 if (l + 1) in cfg.skip_in else dims[l + 1]
        out.append((dims[l], o))
    return out


def color_layer_dims(cfg: ColorConfig):
    """fields.py:138-153."""
    d0 = cfg.d_in + cfg.d_feature
    if cfg.multires_view > 0:
        d0 += embed_dim(cfg.multires_view) - 3
    dims = [d0] + [cfg.d_hidden] * cfg.n_layers + [cfg.d_out]
    return [(dims[l], dims[l + 1]) for l in range(len(dims) - 1)]


def relight_layer_dims(cfg: RelightConfig):
    """(in_layer dims, [rl_mlp dims]); fields.py:305-325."""
    d_in = cfg.d_in + (3 if cfg.include_grad else 0)
    if cfg.multires_view > 0:
        d_in += embed_dim(cfg.multires_view) - 3
    layers = []
    for i in range(cfg.n_layers):
        if i == cfg.y_in_layer - 1 and cfg.y_in_layer == cfg.n_layers:
            layers.append((3 + cfg.d_hidden, cfg.d_out))
        elif i == cfg.y_in_layer - 1:
            layers.append((3 + cfg.d_hidden, cfg.d_hidden))
        elif i == cfg.n_layers - 1:
            layers.append((cfg.d_hidden, cfg.d_out))
        else:
            layers.append((cfg.d_hidden, cfg.d_hidden))
    return (d_in, cfg.d_hidden), layers


# --------------------------------------------------------------------------------------
# parameter initialisation (own recipe; the layout/names follow the reference state_dict)
# --------------------------------------------------------------------------------------
def init_params(cfg: RenderConfig, seed: int = 0, dtype=torch.float32, trained_like: bool = False) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (torch CPU generator).  NOT the reference's RNG stream:
    goldens store either the weights themselves or this recipe's seed + a checksum.

    trained_like=False: geometric-style SDF init (sphere), default-ish colour/relight.
    trained_like=True : SURVEY 8(d) 'W-b': sharper variance, bigger sphere, non-degenerate colour nets.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)

    def randn(*shape, std=1.0, mean=0.0):
        return (torch.randn(*shape, generator=g, dtype=torch.float64) * std + mean)

    P: Dict[str, torch.Tensor] = {}
    d0 = embed_dim(cfg.sdf.multires, cfg.sdf.d_in)
    dims = sdf_layer_dims(cfg.sdf)
    nl = len(dims)
    for l, (i, o) in enumerate(dims):
        if l == nl - 1:
            w = randn(o, i, std=1e-4, mean=math.sqrt(math.pi) / math.sqrt(i))
            b = torch.full((o,), -cfg.sdf.bias, dtype=torch.float64)
            if trained_like:
                w[1:] = randn(o - 1, i, std=1.0 / math.sqrt(i))
                b[0] = -1.5
        elif l == 0 and cfg.sdf.multires > 0:
            w = torch.zeros(o, i, dtype=torch.float64)
            w[:, :3] = randn(o, 3, std=math.sqrt(2) / math.sqrt(o))
            if trained_like:
                for k in range(cfg.sdf.multires):   # mild high-frequency detail, ~1/f spectrum
                    w[:, 3 + 6 * k: 9 + 6 * k] = randn(o, 6, std=0.01 / 2.0 ** k)
            b = torch.zeros(o, dtype=torch.float64)
        elif l in cfg.sdf.skip_in and cfg.sdf.multires > 0:
            w = randn(o, i, std=math.sqrt(2) / math.sqrt(o))
            w[:, -(d0 - 3):] = 0.0
            if trained_like:
                for k in range(cfg.sdf.multires):
                    w[:, i - d0 + 3 + 6 * k: i - d0 + 9 + 6 * k] = randn(o, 6, std=0.01 / 2.0 ** k)
            b = torch.zeros(o, dtype=torch.float64)
        else:
            w = randn(o, i, std=math.sqrt(2) / math.sqrt(o))
            b = torch.zeros(o, dtype=torch.float64) if not trained_like else randn(o, std=0.01)
        if cfg.sdf.weight_norm:
            # perturb g so that g != ||v|| (a freshly wrapped layer has g == ||v||)
            nrm = w.norm(dim=1, keepdim=True)
            P[f"sdf_network.lin{l}.weight_g"] = nrm * (1.0 + 0.05 * randn(o, 1)) if trained_like else nrm.clone()
            P[f"sdf_network.lin{l}.weight_v"] = w
        else:
            P[f"sdf_network.lin{l}.weight"] = w
        P[f"sdf_network.lin{l}.bias"] = b
    P["deviation_network.variance"] = torch.tensor(0.65 if trained_like else cfg.init_val, dtype=torch.float64)
    for l, (i, o) in enumerate(color_layer_dims(cfg.color)):
        w = randn(o, i, std=1.0 / math.sqrt(i))
        b = randn(o, std=0.1)
        if cfg.color.weight_norm:
            P[f"color_network.lin{l}.weight_g"] = w.norm(dim=1, keepdim=True) * (1.0 + 0.05 * randn(o, 1))
            P[f"color_network.lin{l}.weight_v"] = w
        else:
            P[f"color_network.lin{l}.weight"] = w
        P[f"color_network.lin{l}.bias"] = b
    if cfg.relight is not None:
        (i, o), layers = relight_layer_dims(cfg.relight)
        P["relight_network.in_layer.weight"] = randn(o, i, std=1.0 / math.sqrt(i))
        P["relight_network.in_layer.bias"] = randn(o, std=0.1)
        for k, (i, o) in enumerate(layers):
            P[f"relight_network.rl_mlp.{k}.weight"] = randn(o, i, std=1.0 / math.sqrt(i))
            P[f"relight_network.rl_mlp.{k}.bias"] = randn(o, std=0.1)
    return {k: v.to(dtype) for k, v in P.items()}


def init_nerf_params(seed: int = 0, D=8, W=256, multires=10, multires_view=4, skips=(4,), dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights of the NeRF++ background network (fields.py:192-274 built with its defaults, NeuS.py:87-91),
    keyed by the reference's state_dict names ``nerf.*``.  Own recipe (uniform +-1/sqrt(fan_in) from a torch CPU generator), so that
    fixtures need a seed and a checksum instead of 600 k stored floats."""
    g = torch.Generator().manual_seed(seed)
    ch, ch_view = 4 * (1 + 2 * multires), 3 * (1 + 2 * multires_view)
    P = {}

    def lin(name, i, o):
        b = 1.0 / math.sqrt(i)
        P[f"nerf.{name}.weight"] = ((torch.rand(o, i, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)
        P[f"nerf.{name}.bias"] = ((torch.rand(o, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)
    lin("pts_linears.0", ch, W)
    for i in range(D - 1):
        lin(f"pts_linears.{i + 1}", W + ch if i in skips else W, W)
    lin("views_linears.0", ch_view + W, W // 2)
    lin("feature_linear", W, W)
    lin("alpha_linear", W, 1)
    lin("rgb_linear", W // 2, 3)
    return P


def params_checksum(P: Dict[str, torch.Tensor]) -> float:
    s = 0.0
    for k in sorted(P):
        s += float(P[k].double().abs().sum())
    return s


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def positional_encoding(x: torch.Tensor, multires: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]; PositionEncoding.py:51-76."""
    if multires <= 0:
        return x
    outs = [x]
    for k in range(multires):
        f = float(2.0 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, dim=-1)


def effective_weight(P, prefix: str, weight_norm: bool) -> torch.Tensor:
    """nn.utils.weight_norm(dim=0): W = g * v / ||v||_row; fields.py:72-73."""
    if weight_norm:
        v = P[prefix + ".weight_v"]
        g = P[prefix + ".weight_g"]
        return v * (g / v.norm(dim=1, keepdim=True))
    return P[prefix + ".weight"]


def softplus100(x):
    """nn.Softplus(beta=100) (threshold 20); fields.py:77."""
    return F.softplus(x, beta=100.0, threshold=20.0)


def inverse_sigmoid(x, eps=1e-5):
    """lib/utils/transform.py:304-320."""
    x = x.clamp(0.0, 1.0)
    return torch.log(x.clamp(min=eps) / (1.0 - x).clamp(min=eps))


def near_far_from_sphere(rays_o, rays_d):
    """ray_utils.py:7-13."""
    a = (rays_d * rays_d).sum(-1)
    b = 2.0 * (rays_o * rays_d).sum(-1)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


# --------------------------------------------------------------------------------------
# SDF network: value, features, analytic input gradient          (fields.py:81-115)
# --------------------------------------------------------------------------------------
def sdf_forward(P, cfg: SDFConfig, x: torch.Tensor, want_grad: bool = False):
    """Returns (sdf (N,1), feat (N,d_out-1), grad (N,3) or None)."""
    nl = cfg.n_layers + 1
    W = [effective_weight(P, f"sdf_network.lin{l}", cfg.weight_norm) for l in range(nl)]
    B = [P[f"sdf_network.lin{l}.bias"] for l in range(nl)]
    x0 = x * cfg.scale
    e = positional_encoding(x0, cfg.multires)
    h = e
    zs = []
    for l in range(nl):
        if l in cfg.skip_in:
            h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
        z = h @ W[l].t() + B[l]
        zs.append(z)
        h = softplus100(z) if l < nl - 1 else z
    sdf = h[:, :1] / cfg.scale
    feat = h[:, 1:]
    if not want_grad:
        return sdf, feat, None
    # reverse sweep: cotangent of `sdf` w.r.t. z_last is e_0 / scale
    v = (W[nl - 1][0:1, :] / cfg.scale).expand(x.shape[0], -1)   # cotangent of h_{nl-2} (post-activation / concat)
    ce = torch.zeros_like(e)
    for l in range(nl - 2, -1, -1):
        if (l + 1) in cfg.skip_in:
            d_h = W[l].shape[0]
            ce = ce + v[:, d_h:] / math.sqrt(2.0)
            v = v[:, :d_h] / math.sqrt(2.0)
        u = torch.sigmoid(100.0 * zs[l]) * v
        u = torch.where(zs[l] * 100.0 > 20.0, v, u)
        v = u @ W[l]
    if 0 in cfg.skip_in:
        raise NotImplementedError("skip at layer 0")
    ce = ce + v
    # PE Jacobian transpose
    g = ce[:, :3].clone() if cfg.multires > 0 else ce
    for k in range(cfg.multires):
        f = float(2.0 ** k)
        cs = ce[:, 3 + 6 * k: 6 + 6 * k]
        cc = ce[:, 6 + 6 * k: 9 + 6 * k]
        g = g + f * (torch.cos(x0 * f) * cs - torch.sin(x0 * f) * cc)
    return sdf, feat, g * cfg.scale


def sdf_value(P, cfg: SDFConfig, x):
    return sdf_forward(P, cfg, x)[0]


# --------------------------------------------------------------------------------------
# colour + relight networks                                    (fields.py:161-188, 332-368)
# --------------------------------------------------------------------------------------
def color_forward(P, cfg: ColorConfig, pts, normals, view_dirs, feat):
    if cfg.multires_view > 0:
        view_dirs = positional_encoding(view_dirs, cfg.multires_view)
    if cfg.mode == "idr":
        x = torch.cat([pts, view_dirs, normals, feat], dim=-1)
    elif cfg.mode == "no_view_dir":
        x = torch.cat([pts, normals, feat], dim=-1)
    elif cfg.mode == "no_normal":
        x = torch.cat([pts, view_dirs, feat], dim=-1)
    else:
        raise ValueError(cfg.mode)
    nl = cfg.n_layers + 1
    for l in range(nl):
        x = x @ effective_weight(P, f"color_network.lin{l}", cfg.weight_norm).t() + P[f"color_network.lin{l}.bias"]
        if l < nl - 1:
            x = torch.relu(x)
    return torch.sigmoid(x) if cfg.squeeze_out else x


def relight_forward(P, cfg: RelightConfig, rgb, pts, dirs, gradients):
    if cfg.multires_view > 0:
        dirs = positional_encoding(dirs, cfg.multires_view)
    parts = [pts, dirs] + ([gradients] if cfg.include_grad else [])
    h = torch.cat(parts, dim=-1) @ P["relight_network.in_layer.weight"].t() + P["relight_network.in_layer.bias"]
    for i in range(cfg.n_layers):
        h = torch.relu(h)
        if i == cfg.y_in_layer - 1:
            h = torch.cat([rgb, h], dim=-1)
        h = h @ P[f"relight_network.rl_mlp.{i}.weight"].t() + P[f"relight_network.rl_mlp.{i}.bias"]
    if cfg.inv_sigmoid:
        return torch.sigmoid(inverse_sigmoid(rgb) + h), h
    return torch.clamp(rgb + torch.sigmoid(h) - 0.5, 0.0, 1.0), h


# --------------------------------------------------------------------------------------
# sampler                                       (NeuS.py:136-197, ray_utils.py:123-154)
# --------------------------------------------------------------------------------------
def sample_pdf_det(bins, weights, n_samples, u=None):
    """ray_utils.sample_pdf (ray_utils.py:123-154).  u = None: det=True (the render path, NeuS.py:180); u given [..., n_samples]: the draws of
    det=False (ray_utils.py:135-136: torch.rand of that shape on the CPU generator)."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, dtype=cdf.dtype, device=cdf.device)
        u = u.expand(*cdf.shape[:-1], n_samples)
    u = u.to(cdf.dtype).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = (idx - 1).clamp(min=0)
    hi = idx.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b0, b1 = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def exclusive_transmittance(alpha):
    """cumprod([1, 1-a+1e-7])[:-1]"""
    ones = torch.ones_like(alpha[..., :1])
    return torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-7], -1), -1)[..., :-1]


def up_sample(rays_o, rays_d, z, sdf, n_importance, inv_s):
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    rad = torch.linalg.norm(pts, dim=-1)
    inside = (rad[:, :-1] < 1.0) | (rad[:, 1:] < 1.0)
    s0, s1 = sdf[:, :-1], sdf[:, 1:]
    z0, z1 = z[:, :-1], z[:, 1:]
    mid = (s0 + s1) * 0.5
    cos = (s1 - s0) / (z1 - z0 + 1e-5)
    prev = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clamp(-1e3, 0.0) * inside
    dist = z1 - z0
    pc = torch.sigmoid((mid - cos * dist * 0.5) * inv_s)
    nc = torch.sigmoid((mid + cos * dist * 0.5) * inv_s)
    alpha = (pc - nc + 1e-5) / (pc + 1e-5)
    w = alpha * exclusive_transmittance(alpha)
    return sample_pdf_det(z, w, n_importance)


def merge_z(P, cfg: RenderConfig, rays_o, rays_d, z, new_z, sdf, last):
    R, n = z.shape
    zc, idx = torch.sort(torch.cat([z, new_z], -1), dim=-1)
    if not last:
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., None]
        ns = sdf_value(P, cfg.sdf, pts.reshape(-1, 3)).reshape(R, -1)
        sdf = torch.gather(torch.cat([sdf, ns], -1), -1, idx)
    return zc, sdf


def sample_z(P, cfg: RenderConfig, rays_o, rays_d, near, far, t_rand=None):
    """Coarse + hierarchical z (NeuS.py:309-357).  t_rand: (R,1) uniform [0,1) jitter draw or None (no perturb)."""
    S = cfg.n_samples
    lin = torch.linspace(0.0, 1.0, S, dtype=rays_o.dtype, device=rays_o.device)
    z = near[:, None] + (far[:, None] - near[:, None]) * lin[None, :]
    if t_rand is not None:
        z = z + (t_rand - 0.5) * 2.0 / S
    if cfg.n_importance > 0:
        with torch.no_grad():
            R = z.shape[0]
            pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
            sdf = sdf_value(P, cfg.sdf, pts.reshape(-1, 3)).reshape(R, S)
            K = cfg.up_sample_steps
            for i in range(K):
                nz = up_sample(rays_o, rays_d, z, sdf, cfg.n_importance // K, 64 * 2 ** i)
                z, sdf = merge_z(P, cfg, rays_o, rays_d, z, nz, sdf, last=(i + 1 == K))
            z = z.detach()
    return z


# --------------------------------------------------------------------------------------
# render core + full forward           (Color_NeuS.py:24-138, NeuS.py:199-292, 294-408)
# --------------------------------------------------------------------------------------
def sdf_gradient_autograd(P, cfg: SDFConfig, x):
    """SDFNetwork.gradient exactly as the reference executes it (fields.py:105-115): a SECOND forward of the SDF network on x with
    requires_grad and torch.autograd.grad(create_graph=True) through it -- the double-backward graph the reference trains through.
    Same value as the analytic reverse sweep of sdf_forward(want_grad=True); used where the reference's executed WORK is to be timed."""
    with torch.enable_grad():
        xg = x.detach().requires_grad_(True) if not x.requires_grad else x
        y = sdf_forward(P, cfg, xg)[0]
        g = torch.autograd.grad(outputs=y, inputs=xg, grad_outputs=torch.ones_like(y), create_graph=True, retain_graph=True,
                                only_inputs=True)[0]
    return g


def render_core(P, cfg: RenderConfig, rays_o, rays_d, z, cos_anneal_ratio=0.0, background_rgb=None, reference_ops=False):
    R, M = z.shape
    sample_dist = 2.0 / cfg.n_samples
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], sample_dist)], -1)
    mid = z + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(R, M, 3).reshape(-1, 3)
    if reference_ops:   # op for op what the reference runs: forward for (sdf, features), second forward + autograd.grad for the normals
        sdf, feat, _ = sdf_forward(P, cfg.sdf, pts)
        g = sdf_gradient_autograd(P, cfg.sdf, pts)
    else:
        sdf, feat, g = sdf_forward(P, cfg.sdf, pts, want_grad=True)
    out = {}
    if cfg.type == "Color_NeuS":
        gcol = color_forward(P, cfg.color, pts, g, dirs, feat)
        relit, drgb = relight_forward(P, cfg.relight, gcol, pts, dirs, g)
        sampled = relit.reshape(R, M, 3)
    else:
        sampled = color_forward(P, cfg.color, pts, g, dirs, feat).reshape(R, M, 3)
    inv_s = torch.exp(P["deviation_network.variance"] * 10.0).clamp(1e-6, 1e6)
    tc = (dirs * g).sum(-1, keepdim=True)
    ic = -(torch.relu(-tc * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + torch.relu(-tc) * cos_anneal_ratio)
    d_ = dists.reshape(-1, 1)
    pc = torch.sigmoid((sdf - ic * d_ * 0.5) * inv_s)
    nc = torch.sigmoid((sdf + ic * d_ * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).reshape(R, M).clamp(0.0, 1.0)
    pn = torch.linalg.norm(pts, dim=-1).reshape(R, M)
    inside = (pn < 1.0).to(z.dtype)
    relax = (pn < 1.2).to(z.dtype)
    w = alpha * exclusive_transmittance(alpha)
    wsum = w.sum(-1, keepdim=True)
    color = (sampled * w[..., None]).sum(1)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - wsum)
    gn = torch.linalg.norm(g.reshape(R, M, 3), dim=-1)
    gerr = (relax * (gn - 1.0) ** 2).sum() / (relax.sum() + 1e-5)
    out.update(color_fine=color, s_val=(1.0 / inv_s).expand(R, M).mean(-1, keepdim=True), cdf_fine=pc.reshape(R, M),
               weight_sum=wsum, weight_max=w.max(-1, keepdim=True)[0], gradients=g.reshape(R, M, 3), weights=w,
               gradient_error=gerr, inside_sphere=inside, depth=(w * z).sum(-1))
    if cfg.type == "Color_NeuS":
        out["global_color"] = (gcol.reshape(R, M, 3) * w[..., None]).sum(1)
        out["delta_relight"] = drgb.reshape(R, M, 3)
    return out


def render(P, cfg: RenderConfig, rays_o, rays_d, near, far, t_rand=None, cos_anneal_ratio=0.0,
           background_rgb=None, z_vals=None, reference_ops=False):
    """Full forward.  t_rand (R,1) is the uniform draw the reference takes from torch.rand([R,1]) on the CPU
    generator (NeuS.py:325); None means perturb == 0.  z_vals overrides the sampler (parity gate G2)."""
    if cfg.n_outside > 0:
        raise NotImplementedError("N_OUTSIDE > 0 (NeRF++ background) is outside the hot path")
    z = sample_z(P, cfg, rays_o, rays_d, near, far, t_rand) if z_vals is None else z_vals
    out = render_core(P, cfg, rays_o, rays_d, z, cos_anneal_ratio, background_rgb, reference_ops=reference_ops)
    out["z_vals"] = z
    return out


# --------------------------------------------------------------------------------------
# the consumer of the path: loss                         (NeuS_Trainer.py:129-171)
# --------------------------------------------------------------------------------------
def compute_loss(out, rgb_gt, mask=None, lambda_fine=1.0, lambda_eikonal=0.1, lambda_mask=0.1,
                 lambda_relight=1.0, rgb_loss="mse", include_mask=True):
    rgb = F.mse_loss(out["color_fine"], rgb_gt) if rgb_loss == "mse" else F.l1_loss(out["color_fine"], rgb_gt)
    loss = lambda_fine * rgb + lambda_eikonal * out["gradient_error"]
    parts = dict(rgb_fine_loss=rgb, eikonal_loss=out["gradient_error"])
    if lambda_mask != 0 and mask is not None:
        ws = out["weight_sum"].squeeze(-1).clamp(1e-3, 1.0 - 1e-3)
        ml = F.binary_cross_entropy(ws, mask)
        loss = loss + lambda_mask * ml
        parts["mask_loss"] = ml
    if lambda_relight != 0 and "delta_relight" in out:
        dr = out["delta_relight"]
        if include_mask and mask is not None:
            dr = dr * mask[:, None, None]
        rl = dr.mean() ** 2
        loss = loss + lambda_relight * rl
        parts["relight_loss"] = rl
    parts["loss"] = loss
    return loss, parts


def tiny_config() -> RenderConfig:
    """BASELINE config C1 / SURVEY D9: 64-wide 2-layer MLPs, S=16 + I=16."""
    return RenderConfig(type="Color_NeuS", n_samples=16, n_importance=16, up_sample_steps=4, perturb=1.0,
                        sdf=SDFConfig(d_out=65, d_hidden=64, n_layers=2, skip_in=[]),
                        color=ColorConfig(d_feature=64, mode="no_view_dir", d_in=6, d_hidden=64, n_layers=2,
                                          multires_view=0),
                        relight=RelightConfig(d_hidden=64, n_layers=2, y_in_layer=1))


def dtu_config(n_samples=64, n_importance=64) -> RenderConfig:
    """config/Color_NeuS_dtu.yml:23-60 renderer block."""
    return RenderConfig(type="Color_NeuS", n_samples=n_samples, n_importance=n_importance,
                        color=ColorConfig(mode="no_view_dir", d_in=6, multires_view=0))
