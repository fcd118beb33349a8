"""Ray tools of the caller side of the render path: drop-in counterparts of
lib/models/tools/ray_utils.py `get_rays_multicam` (:16-87), `get_rays_at` (:90-119), `near_far_from_sphere` (:7-13).

The reference's `get_rays_multicam` materialises directions/origins for ALL N*H*W pixels (~184 MB per step at DTU size) and then
gathers n_rays of them.  Here the random pixel selection is done first -- with the SAME torch RNG calls in the same order, so a
seeded run picks identical pixels -- and rays are built only for the selected pixels.  All arithmetic per selected ray is the
reference's formula, kept in differentiable torch ops (n_rays x 3 elements: negligible work) so that learnable poses / focal
(config/Color_NeuS_iho.yml:18-20) still receive gradients through rays_o / rays_d."""
import torch


def near_far_from_sphere(rays_o, rays_d):
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return (mid - 1.0).squeeze(), (mid + 1.0).squeeze()


def _pixel_dirs(px, py, focal, H, W, normalize, opengl):
    y, z = (-1, -1) if opengl else (1, 1)
    dirs = torch.stack([(px - W * 0.5) / focal[0], y * (py - H * 0.5) / focal[1], z * torch.ones_like(px)], -1)
    if normalize:
        dirs = dirs / torch.norm(dirs, dim=-1).unsqueeze(-1)
    return dirs


def get_rays_multicam(c2w, focal, image, n_rays, normalize=False, mask=None, mask_rate=0.9, return_mask=False, opengl=False):
    """Random n rays in world space from N cameras; same signature, return values and RNG consumption as the reference."""
    assert c2w.dim() == 3 and image.dim() == 4, "this is a multicam implementation"
    device = c2w.device
    H, W = image.shape[1], image.shape[2]
    if mask is None:
        rays_idx_raw = torch.randint(0, H * W, (n_rays,)).to(device)
        mask_all = None
    else:
        mask_all = mask.reshape(-1)
        valide_index = torch.where(mask_all > 0)[0]
        rand_valid_index = torch.randperm(valide_index.shape[0])
        n_rays_in_mask = int(mask_rate * n_rays)
        if n_rays_in_mask > valide_index.shape[0]:
            n_rays_in_mask = valide_index.shape[0]
        n_rays_in_bkg = n_rays - n_rays_in_mask
        invalid_index = torch.where(mask_all == 0)[0]
        rand_invalid_index = torch.randperm(invalid_index.shape[0])
        rays_idx_raw = torch.cat([valide_index[rand_valid_index[:n_rays_in_mask]], invalid_index[rand_invalid_index[:n_rays_in_bkg]]], dim=-1)
        rays_idx_raw = rays_idx_raw[torch.randperm(rays_idx_raw.shape[0])]
    cam = torch.div(rays_idx_raw, H * W, rounding_mode="floor")
    pix = rays_idx_raw - cam * (H * W)
    py = torch.div(pix, W, rounding_mode="floor").to(torch.float32)
    px = (pix - torch.div(pix, W, rounding_mode="floor") * W).to(torch.float32)
    dirs = _pixel_dirs(px, py, focal, H, W, normalize, opengl)                       # [n_rays, 3]
    rot = c2w[cam, :3, :3]                                                            # [n_rays, 3, 3]
    rays_d = torch.sum(dirs[:, None, :] * rot, -1)
    rays_o = c2w[cam, :3, -1]
    rgb = image.reshape(-1, 3)[rays_idx_raw]
    if return_mask:
        assert mask is not None
        return rays_o, rays_d, rgb, mask_all[rays_idx_raw]
    return rays_o, rays_d, rgb, None


def get_rays_at(c2w, focal, H, W, normalize=False, opengl=False):
    """All rays of one camera, [H, W, 3] each (reference ray_utils.py:90-119)."""
    assert c2w.dim() == 2, "this is a sigle camera implementation"
    device = c2w.device
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="xy")
    dirs = _pixel_dirs(i.to(device), j.to(device), focal, H, W, normalize, opengl)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d
