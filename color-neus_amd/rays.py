"""Ray tools of the caller side of the render path: drop-in counterparts of the reference's
``get_rays_multicam`` / ``get_rays_at`` / ``near_far_from_sphere`` (lib/models/tools/ray_utils.py:7-119) on the device library.

The reference builds directions and origins for ALL N*H*W pixels of the batch (about 184 MB per step at DTU size) and gathers
n_rays of them.  Here the pixels are chosen first and one kernel (cnr_gen_rays) builds rays, colours and mask values only for
those; ``rays_for_training`` also folds in what NeuS_Trainer.render does next (origin / radius normalisation, near / far).
Learnable poses and focal lengths (config/Color_NeuS_iho.yml:18-20) get their gradients from cnr_gen_rays_backward.

Pixel choice consumes the torch CPU generator exactly like the reference (same calls, same order), so a seeded run picks the same
pixels: that is the only part left in torch -- it IS the reference's random stream."""
import ctypes as C
import os
import warnings

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream_of(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else C.c_void_p(0)


def _library(library):
    return library if isinstance(library, _lib.RenderLibrary) else _lib.load_library(library)


def choose_pixels(n_rays, pixels_per_image, device, mask=None, mask_rate=0.9):
    """Flat pixel indices (camera * H * W + row * W + column) of one training batch.

    Without a mask: uniform draws over ONE image's pixels, as the reference does (its indices never leave camera 0,
    ray_utils.py:58).  With a mask: a share ``mask_rate`` of the batch from the foreground pixels, the rest from the background,
    both without replacement, then shuffled.  Random-number consumption: randint | randperm(#fg), randperm(#bg), randperm(n)."""
    if mask is None:
        return torch.randint(0, pixels_per_image, (n_rays,)).to(device)
    flat = mask.reshape(-1)
    fg = torch.nonzero(flat > 0, as_tuple=True)[0]
    fg_order = torch.randperm(fg.shape[0])
    want_fg = int(mask_rate * n_rays)
    if want_fg > fg.shape[0]:
        warnings.warn(f"only {fg.shape[0]} foreground pixels for {want_fg} requested rays")
        want_fg = fg.shape[0]
    bg = torch.nonzero(flat == 0, as_tuple=True)[0]
    bg_order = torch.randperm(bg.shape[0])
    chosen = torch.cat([fg[fg_order[:want_fg]], bg[bg_order[:n_rays - want_fg]]], dim=-1)
    return chosen[torch.randperm(chosen.shape[0])]


_CHECK_INDICES = os.environ.get("CNR_CHECK_INDICES", "0") not in ("", "0")
_BAD = {}   # device -> int32 counter written by the ray kernel: pixel indices outside [0, n_cams * H * W) since the last check


def _bad_counter(dev):
    t = _BAD.get(dev)
    if t is None:
        t = _BAD[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def bad_index_count(device=None):
    """Out-of-range pixel indices the ray kernel has met on ``device`` (default: every device used) since the last call; resets the counter.
    Reads device memory, i.e. synchronises -- call it where the step synchronises anyway (next to the loss' .item())."""
    n = 0
    for dev, t in _BAD.items():
        if device is None or torch.device(device) == dev:
            n += int(t.item())
            t.zero_()
    return n


def raise_if_bad_indices(device=None):
    """The reference's torch indexing raises IndexError on an out-of-range pixel index (ray_utils.py:63-76); the ray kernel cannot raise -- it
    makes that ray NaN and counts it.  Call this at the step's synchronisation point (before the optimiser step, if NaN gradients must not
    reach the optimiser state): raises IndexError like the reference, one step late at most."""
    n = bad_index_count(device)
    if n:
        raise IndexError(f"{n} pixel index/indices outside [0, n_cams * H * W) reached cnr_gen_rays since the last check (their rays are NaN)")


class _GenRays(torch.autograd.Function):
    """autograd edge around cnr_gen_rays / cnr_gen_rays_backward."""

    @staticmethod
    def forward(ctx, lib, pix_idx, n, c2w, focal, H, W, normalize, opengl, image, mask, origin, radius, want_nearfar):
        dev = c2w.device
        c2w_c = c2w.detach().reshape(-1, 4, 4).contiguous().float()
        focal_c = focal.detach().reshape(-1).contiguous().float().to(dev)
        # every operand is dereferenced on c2w's device.  The image / mask stacks must already live there: moving a host-resident stack
        # (about 1 GB for DTU) on every step would be a silent per-step upload, so that is an error like the device mismatch the reference's
        # torch ops would raise; the small operands (focal, origin, the index list) are moved.
        for name, t in (("image", image), ("mask", mask)):
            if t is not None and t.device != dev:
                raise ValueError(f"{name} is on {t.device} but c2w is on {dev}: keep the image / mask stacks on the device that generates the rays")
        img = image.detach().contiguous().float() if image is not None else None
        msk = mask.detach().contiguous().float() if mask is not None else None
        org = origin.detach().reshape(-1).contiguous().float().to(dev) if origin is not None else None
        idx = pix_idx.to(dev).contiguous().to(torch.int64) if pix_idx is not None else None
        n_cams = c2w_c.shape[0]
        for name, t in (("image", img), ("mask", msk)):
            if t is not None and t.shape[0] != n_cams:
                raise ValueError(f"{name} holds {t.shape[0]} cameras but c2w holds {n_cams}")
        # range of the indices: the ray kernel checks every index on the device (an index outside [0, n_cams * H * W) reads nothing and makes
        # that ray's outputs NaN; its backward contributes nothing) -- no host cost, no out-of-bounds access.  The host-side check below
        # raises like the reference's torch indexing would, but costs two device-to-host syncs: on request only (CNR_CHECK_INDICES=1);
        # the indices of choose_pixels, the only producer inside this module, are in range by construction
        if idx is not None and idx.numel() > 0 and _CHECK_INDICES:
            lo, hi = int(idx.min()), int(idx.max())
            if lo < 0 or hi >= n_cams * H * W:
                raise IndexError(f"pixel index range [{lo}, {hi}] outside [0, {n_cams * H * W})")
        f32 = dict(dtype=torch.float32, device=dev)
        rays_o, rays_d = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
        rgb = torch.empty(n, 3, **f32) if img is not None else None
        msel = torch.empty(n, **f32) if msk is not None else None
        near = torch.empty(n, **f32) if want_nearfar else None
        far = torch.empty(n, **f32) if want_nearfar else None
        rc = lib.lib.cnr_gen_rays(_ptr(idx), n, _ptr(c2w_c), c2w_c.shape[0], _ptr(focal_c), H, W, int(normalize), int(opengl), _ptr(img), _ptr(msk),
                                  _ptr(org), float(radius), _ptr(rays_o), _ptr(rays_d), _ptr(rgb), _ptr(msel), _ptr(near), _ptr(far),
                                  _ptr(_bad_counter(dev)) if idx is not None else C.c_void_p(0), _stream_of(c2w_c))
        lib.check(rc, "cnr_gen_rays")
        ctx.lib, ctx.meta = lib, (n, H, W, int(normalize), int(opengl), float(radius), c2w.shape, focal.shape)
        ctx.save_for_backward(idx if idx is not None else torch.empty(0, dtype=torch.int64, device=dev), c2w_c, focal_c,
                              org if org is not None else torch.empty(0, device=dev))
        outs = [rays_o, rays_d]
        nd = [t for t in (rgb, msel) if t is not None]
        ctx.mark_non_differentiable(*nd)
        ctx.n_extra = (rgb is not None, msel is not None, want_nearfar)
        return tuple(outs + [rgb if rgb is not None else torch.empty(0, device=dev), msel if msel is not None else torch.empty(0, device=dev),
                             near if near is not None else torch.empty(0, device=dev), far if far is not None else torch.empty(0, device=dev)])

    @staticmethod
    def backward(ctx, d_o, d_d, _d_rgb, _d_mask, d_near, d_far):
        idx, c2w_c, focal_c, org = ctx.saved_tensors
        n, H, W, normalize, opengl, radius, c2w_shape, focal_shape = ctx.meta
        dev = c2w_c.device
        need = ctx.needs_input_grad
        if not (need[3] or need[4]):
            return (None,) * 14
        z3 = lambda g: g.contiguous().float() if g is not None else None
        d_o, d_d = z3(d_o), z3(d_d)
        has_nf = ctx.n_extra[2] and d_near is not None and d_far is not None and d_near.numel() == n
        d_near = z3(d_near) if has_nf else None
        d_far = z3(d_far) if has_nf else None
        d_c2w = torch.empty_like(c2w_c)
        d_focal = torch.empty(2, dtype=torch.float32, device=dev)
        scratch = torch.empty(c2w_c.shape[0] * 2, dtype=torch.float32, device=dev)
        rc = ctx.lib.lib.cnr_gen_rays_backward(_ptr(idx if idx.numel() else None), n, _ptr(c2w_c), c2w_c.shape[0], _ptr(focal_c), H, W, normalize, opengl,
                                               _ptr(org if org.numel() else None), radius, _ptr(d_o), _ptr(d_d), _ptr(d_near), _ptr(d_far),
                                               _ptr(d_c2w), _ptr(d_focal), _ptr(scratch), scratch.numel() * 4, _stream_of(c2w_c))
        ctx.lib.check(rc, "cnr_gen_rays_backward")
        return (None, None, None, d_c2w.reshape(c2w_shape) if need[3] else None, d_focal.reshape(focal_shape) if need[4] else None,
                None, None, None, None, None, None, None, None, None)


def _generate(lib, pix_idx, n, c2w, focal, H, W, normalize, opengl, image=None, mask=None, origin=None, radius=1.0, want_nearfar=False):
    o, d, rgb, msel, near, far = _GenRays.apply(lib, pix_idx, n, c2w, focal, H, W, normalize, opengl, image, mask, origin, radius, want_nearfar)
    return o, d, (rgb if image is not None else None), (msel if mask is not None else None), (near if want_nearfar else None), (far if want_nearfar else None)


def get_rays_multicam(c2w, focal, image, n_rays, normalize=False, mask=None, mask_rate=0.9, return_mask=False, opengl=False, library=None):
    """Random n rays in world space from N cameras: the reference's signature and return values (ray_utils.py:16-87)."""
    assert c2w.dim() == 3 and image.dim() == 4, "c2w [N,4,4] and image [N,H,W,3] expected (multi-camera form)"
    H, W = image.shape[1], image.shape[2]
    idx = choose_pixels(n_rays, H * W, c2w.device, mask, mask_rate)
    if return_mask:
        assert mask is not None
    o, d, rgb, msel, _, _ = _generate(_library(library), idx, idx.shape[0], c2w, focal, H, W, normalize, opengl, image=image,
                                      mask=mask if return_mask else None)
    return o, d, rgb, msel


def get_rays_at(c2w, focal, H, W, normalize=False, opengl=False, library=None):
    """All rays of one camera, [H, W, 3] each (ray_utils.py:90-119)."""
    assert c2w.dim() == 2, "c2w [4,4] expected (single-camera form)"
    o, d, _, _, _, _ = _generate(_library(library), None, H * W, c2w, focal, H, W, normalize, opengl)
    return o.reshape(H, W, 3), d.reshape(H, W, 3)


def rays_for_training(c2w, focal, image, n_rays, origin, radius, normalize=False, mask=None, mask_rate=0.9, return_mask=False, opengl=False,
                      library=None):
    """What NeuS_Trainer.render does in front of the renderer call (NeuS_Trainer.py:104-120) in one launch: pixel choice, rays,
    (rays_o - origin) / radius, near / far from the unit sphere, colours and mask values of the chosen pixels.
    Returns rays_o, rays_d, near, far, rgb_gt, mask_select (None unless return_mask)."""
    assert c2w.dim() == 3 and image.dim() == 4
    H, W = image.shape[1], image.shape[2]
    idx = choose_pixels(n_rays, H * W, c2w.device, mask, mask_rate)
    o, d, rgb, msel, near, far = _generate(_library(library), idx, idx.shape[0], c2w, focal, H, W, normalize, opengl, image=image,
                                           mask=mask if return_mask else None, origin=torch.as_tensor(origin, dtype=torch.float32), radius=float(radius),
                                           want_nearfar=True)
    return o, d, near, far, rgb, msel


def near_far_from_sphere(rays_o, rays_d):
    """near / far of the unit sphere along each ray (ray_utils.py:7-13); differentiable torch ops on [n, 3] tensors (callers that
    already hold rays; rays_for_training gets the same values from the ray kernel)."""
    a = (rays_d * rays_d).sum(-1)
    mid = -(rays_o * rays_d).sum(-1) / a
    return mid - 1.0, mid + 1.0
