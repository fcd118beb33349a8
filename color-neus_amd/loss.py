"""The consumer of the render path: NeuS_Trainer.compute_loss (lib/models/NeuS_Trainer.py:129-171), needed by
the bench / multi-GPU harness to drive the backward pass with the reference's objective.

``global_stats`` lets a ray-sharded run reproduce the single-GPU objective exactly: the eikonal term is a ratio of
sums over ALL rays (Color_NeuS.py:122-123) and the relight term is the square of a mean over ALL samples
(NeuS_Trainer.py:153); see parallel.py."""
import torch
import torch.nn.functional as F


def compute_loss(render_dict, rgb_gt, mask=None, lambda_fine=1.0, lambda_eikonal=0.1, lambda_mask=0.1, lambda_relight=1.0,
                 rgb_loss_type="mse", include_mask=True):
    rgb = render_dict["color_fine"]
    rgb_fine_loss = F.mse_loss(rgb, rgb_gt) if rgb_loss_type == "mse" else F.l1_loss(rgb, rgb_gt)
    loss = lambda_fine * rgb_fine_loss
    eikonal_loss = render_dict["gradient_error"]
    loss = loss + lambda_eikonal * eikonal_loss
    loss_dict = {"rgb_fine_loss": rgb_fine_loss, "eikonal_loss": eikonal_loss}
    if lambda_mask != 0 and mask is not None:
        mask_loss = F.binary_cross_entropy(render_dict["weight_sum"].squeeze(-1).clip(1e-3, 1.0 - 1e-3), mask)
        loss = loss + lambda_mask * mask_loss
        loss_dict["mask_loss"] = mask_loss
    if lambda_relight != 0 and "delta_relight" in render_dict:
        dr = render_dict["delta_relight"]
        if include_mask and mask is not None:
            dr = dr * mask[:, None, None]
        relight_loss = torch.mean(dr) ** 2
        loss = loss + lambda_relight * relight_loss
        loss_dict["relight_loss"] = relight_loss
    loss_dict["loss"] = loss
    return loss, loss_dict


# ---------------------------------------------------------------------------------------------------------------------
# Fused version: the same objective through two kernels of the render library (cnr_loss_sums /
# cnr_loss_grads, include/colorneus_render.h) instead of ~25 element-wise / reduction launches and their autograd graph.
# Ray-sharded runs pass n_rays_global (+ group): cnr_loss_shard_stats, one 5-float all-reduce, cnr_loss_shard_combine reproduce the
# single-GPU objective exactly (same construction as parallel.sharded_loss); three library launches between forward and backward.
# ---------------------------------------------------------------------------------------------------------------------
import ctypes as _C


def _p(t):
    return _C.c_void_p(t.data_ptr()) if t is not None else _C.c_void_p(0)


def _stream(t):
    return _C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else _C.c_void_p(0)


_SCRATCH = {}   # (library path, device, stream) -> the loss scratch of that stream


def _loss_scratch(lib, dev, R):
    """The scratch of the loss kernels for the current stream of ``dev``.  Its tail holds the completion counter of the one-launch forms
    (include/colorneus_render.h: zero before the first use, left zero by every call), so the buffer is created zeroed ONCE per stream and
    reused: calls on one stream are ordered, calls on different streams get different buffers."""
    nb = lib.lib.cnr_loss_scratch_bytes(R)
    key = (lib.path, str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nb:
        t = torch.zeros(nb, dtype=torch.uint8, device=dev)
        _SCRATCH[key] = t
    return t, nb


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lib, lcfg, lambdas, n_rays_global, group, color, wsum, gerr, eik_sums, drel, gt, mask, n_samples_=0):
        lam_f, lam_e, lam_m, lam_r = lambdas
        ctx.set_materialize_grads(False)   # (the four logging outputs are non-differentiable: no zero tensors for them in backward)
        R = color.shape[0]
        per_ray = drel is not None and drel.dim() == 1    # [R] sums over samples and rgb (renderer training_outputs="loss_only")
        M = (n_samples_ if per_ray else drel.shape[1]) if drel is not None else 1
        dev = color.device
        color_c, gt_c = color.contiguous(), gt.contiguous()
        wsum_c = wsum.reshape(-1).contiguous()
        drel_c = drel.contiguous() if drel is not None else None
        mask_c = mask.contiguous() if mask is not None else None
        scratch, nb = _loss_scratch(lib, dev, R)
        import torch.distributed as dist
        world = dist.get_world_size(group) if (dist.is_initialized() and n_rays_global is not None) else 1
        Rg = float(n_rays_global if n_rays_global is not None else R)
        use_mask, use_rel = (lam_m != 0 and mask is not None), (lam_r != 0 and drel is not None)
        f32 = dict(dtype=torch.float32, device=dev)
        if n_rays_global is None:
            # single process: the two reduction phases and the scalar tail of the objective in ONE launch (cnr_loss_forward)
            sums, out8 = torch.empty(4, **f32), torch.empty(8, **f32)
            gerr_c = gerr.detach().reshape(-1).to(**f32).contiguous()
            lib.check(lib.lib.cnr_loss_forward(_C.byref(lcfg), _p(color_c), _p(wsum_c), _p(drel_c), int(per_ray), _p(gt_c), _p(mask_c), _p(gerr_c), R, int(M), Rg,
                                               int(use_mask), int(use_rel), _p(sums), _p(out8), _p(scratch), nb, _stream(color_c)), "cnr_loss_forward")
            eik_factor = None
        else:
            # ray-sharded: this rank's statistics (one launch), ONE 5-float all-reduce, the scalar tail on the reduced statistics (one launch);
            # no torch arithmetic in between (include/colorneus_render.h, cnr_loss_shard_*)
            stats, out8 = torch.empty(8, **f32), torch.empty(8, **f32)
            eik_c = eik_sums.detach().reshape(-1).to(**f32).contiguous()
            lib.check(lib.lib.cnr_loss_shard_stats(_C.byref(lcfg), _p(color_c), _p(wsum_c), _p(drel_c), int(per_ray), _p(gt_c), _p(mask_c), _p(eik_c), R, int(M),
                                                   _p(stats), _p(scratch), nb, _stream(color_c)), "cnr_loss_shard_stats")
            if world > 1:
                dist.all_reduce(stats[:5], op=dist.ReduceOp.SUM, group=group)
            lib.check(lib.lib.cnr_loss_shard_combine(_C.byref(lcfg), _p(stats), Rg, int(M), int(use_mask), int(use_rel), _p(out8), _stream(color_c)),
                      "cnr_loss_shard_combine")
            eik_factor = out8[6]
        loss, rgb_loss, eik, mask_out, rel_out, mean_rel = out8[:6].unbind(0)
        ctx.lib, ctx.lcfg, ctx.Rg, ctx.M = lib, lcfg, Rg, M
        ctx.has = (use_mask, use_rel, eik_factor is not None)
        ctx.shapes = (color.shape, wsum.shape, tuple(drel.shape) if drel is not None else None)
        ctx.save_for_backward(color_c, wsum_c, gt_c, mask_c if mask_c is not None else torch.empty(0, device=dev), mean_rel,
                              eik_factor if eik_factor is not None else mean_rel)
        # only ``loss`` carries gradients: the four components are logging values (the reference's loss_dict entries are read with
        # .item(), NeuS_Trainer.py:155-171); back-propagating through them would bypass the fused coefficients, so they are all
        # non-differentiable outputs rather than some silently yielding zero gradients
        ctx.mark_non_differentiable(rgb_loss, eik, mask_out, rel_out)
        return loss, rgb_loss, eik, mask_out, rel_out

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        if g_loss is None:   # nothing depends on ``loss`` (cannot happen through the components: they are non-differentiable)
            return (None,) * 13
        color_c, wsum_c, gt_c, mask_c, mean_rel, eik_factor = ctx.saved_tensors
        has_mask, has_rel, has_factor = ctx.has
        lib, lcfg, Rg, M = ctx.lib, ctx.lcfg, ctx.Rg, ctx.M
        dev = color_c.device
        R = color_c.shape[0]
        mask_t = mask_c if mask_c.numel() else None
        # coefficients on the device from the (device-resident) upstream gradient: no host -> device copy, hence no host stall
        g = g_loss.to(torch.float32)
        d_color = torch.empty_like(color_c)
        d_wsum = torch.empty(R, dtype=torch.float32, device=dev)
        mask_g = mask_t if has_mask or lcfg.include_mask else None
        # one launch (cnr_loss_backward): coefficients + gradients; coef[3] = d loss / d gradient_error (incl. the shard's eikonal factor)
        coef = torch.empty(4, dtype=torch.float32, device=dev)
        want_drel = has_rel and ctx.shapes[2] is not None
        # d mean(delta_relight * mask)^2 / d delta_relight[r, j, c] = 2 mean / n * mask[r]: one value per ray, written by the same launch
        per_ray = torch.empty(R, dtype=torch.float32, device=dev) if want_drel else None
        lib.check(lib.lib.cnr_loss_backward(_C.byref(lcfg), _p(color_c), _p(wsum_c), _p(gt_c), _p(mask_g), R, int(M), _p(g.reshape(-1).contiguous()),
                                            _p(mean_rel), _p(eik_factor if has_factor else None), Rg, int(has_mask), int(has_rel), _p(coef), _p(d_color),
                                            _p(d_wsum), _p(per_ray), _stream(color_c)), "cnr_loss_backward")
        d_drel = None
        if want_drel:
            # handed to the renderer's backward as an expanded (stride-0) view -- its compositor backward takes the per-ray vector, no
            # [R][M][3] buffer is written
            d_drel = per_ray.reshape(R, 1, 1).expand(ctx.shapes[2]) if len(ctx.shapes[2]) == 3 else per_ray.reshape(ctx.shapes[2])
        d_gerr = coef[3].reshape(g_loss.shape)
        return (None, None, None, None, None, d_color.reshape(ctx.shapes[0]), d_wsum.reshape(ctx.shapes[1]), d_gerr, None, d_drel, None, None, None)


def compute_loss_fused(render_dict, rgb_gt, mask=None, lambda_fine=1.0, lambda_eikonal=0.1, lambda_mask=0.1, lambda_relight=1.0,
                       rgb_loss_type="mse", include_mask=True, n_rays_global=None, group=None, library=None):
    """Same objective and return convention as compute_loss, evaluated by the render library's loss kernels.  With
    ``n_rays_global`` (ray-sharded data parallel) the returned loss is the GLOBAL value on every rank and its backward yields the
    rank-local gradients whose sum over ranks is the single-GPU gradient."""
    from ._lib import CnrLossConfig, load_library
    lib = library if library is not None else load_library()
    lcfg = CnrLossConfig(lambda_fine, lambda_eikonal, lambda_mask, lambda_relight, 0 if rgb_loss_type == "mse" else 1, 1 if include_mask else 0)
    drel = render_dict.get("delta_relight") if lambda_relight != 0 else None
    n_samples = 0
    if drel is None and lambda_relight != 0 and render_dict.get("delta_relight_ray_sum") is not None:
        # the renderer's training_outputs="loss_only" form: per-ray sums of delta_relight, no [R][M][3] tensor
        drel = render_dict["delta_relight_ray_sum"]
        n_samples = int(render_dict["weights"].shape[1])
    eik_sums = render_dict.get("eik_sums")
    if n_rays_global is not None and eik_sums is None:
        raise ValueError("ray-sharded fused loss needs the renderer's eik_sums output")
    loss, rgb_l, eik_l, mask_l, rel_l = _FusedLoss.apply(lib, lcfg, (lambda_fine, lambda_eikonal, lambda_mask, lambda_relight), n_rays_global,
                                                       group, render_dict["color_fine"], render_dict["weight_sum"],
                                                       render_dict["gradient_error"], eik_sums, drel, rgb_gt, mask, n_samples)
    loss_dict = {"rgb_fine_loss": rgb_l, "eikonal_loss": eik_l, "loss": loss}
    if lambda_mask != 0 and mask is not None:
        loss_dict["mask_loss"] = mask_l
    if drel is not None:
        loss_dict["relight_loss"] = rel_l
    return loss, loss_dict
