"""Synthetic workloads for benches and multi-GPU tests (no dataset exists offline): an 800x800 pinhole view of the unit
sphere and a 'trained-like' weight regime (sharper variance, radius-0.5 sphere)."""
import math

import torch


def synthetic_view(height=800, width=800, focal=1111.1, seed=1, device="cpu"):
    """All H*W rays of one camera on a sphere of radius 2.5-3.0 looking at the origin.  Returns rays_o, rays_d (normalised),
    near, far with the reference's `mid +- 1` rule (ray_utils.py:7-13), rgb targets ~U[0,1), mask ~ Bernoulli(0.7)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = torch.randn(3, generator=g, dtype=torch.float64)
    c = c / c.norm() * (2.5 + 0.5 * float(torch.rand(1, generator=g, dtype=torch.float64)))
    fwd = -c / c.norm()
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    i, j = torch.meshgrid(torch.arange(width, dtype=torch.float64), torch.arange(height, dtype=torch.float64), indexing="xy")
    dirs = ((i - width * 0.5) / focal)[..., None] * right + ((j - height * 0.5) / focal)[..., None] * down + fwd
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    rays_d = dirs.reshape(-1, 3)
    rays_o = c.expand_as(rays_d).contiguous()
    a = (rays_d * rays_d).sum(-1)
    b = 2.0 * (rays_o * rays_d).sum(-1)
    mid = 0.5 * (-b) / a
    n = rays_d.shape[0]
    rgb = torch.rand(n, 3, generator=g)
    mask = (torch.rand(n, generator=g) < 0.7).float()
    f = lambda t: t.float().to(device)
    return f(rays_o), f(rays_d), f(mid - 1.0), f(mid + 1.0), rgb.to(device), mask.to(device)


def synthetic_camera(height=800, width=800, focal=1111.1, seed=1, device="cpu"):
    """The camera of synthetic_view in the form the ray generator takes (rays.rays_for_training / cnr_gen_rays): c2w [1, 4, 4] whose
    columns are right / down / forward / centre, focal [2], image [1, H, W, 3] ~U[0, 1), mask [1, H, W] ~ Bernoulli(0.7) -- the same
    pixel -> ray map (normalised directions, pixel (i, j) -> ((i - W / 2) / f, (j - H / 2) / f, 1)), evaluated per chosen pixel on the device."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = torch.randn(3, generator=g, dtype=torch.float64)
    c = c / c.norm() * (2.5 + 0.5 * float(torch.rand(1, generator=g, dtype=torch.float64)))
    fwd = -c / c.norm()
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, c
    n = height * width
    rgb = torch.rand(n, 3, generator=g)
    mask = (torch.rand(n, generator=g) < 0.7).float()
    return (c2w.float()[None].contiguous().to(device), torch.tensor([focal, focal], dtype=torch.float32, device=device),
            rgb.reshape(1, height, width, 3).contiguous().to(device), mask.reshape(1, height, width).contiguous().to(device))


@torch.no_grad()
def make_trained_like_(renderer, radius=0.5, variance=0.65):
    """Move freshly initialised weights into a 'trained-like' regime: inv_s = exp(10*variance) ~ 665 and an SDF that is a
    sphere of the given radius (geometric init gives sdf ~ |x| - bias/scale)."""
    renderer.deviation_network.variance.fill_(variance)
    top = getattr(renderer.sdf_network, f"lin{renderer.rcfg.sdf_n_layers}")
    top.bias[0] = -radius * renderer.rcfg.sdf_scale
    if hasattr(top, "weight_v"):
        top.weight_v[1:].normal_(0.0, 1.0 / math.sqrt(top.weight_v.shape[1]))
        top.weight_g.copy_(top.weight_v.norm(dim=1, keepdim=True))
    return renderer
