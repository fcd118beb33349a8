"""Ray-sharded data parallelism: one process per GPU, rays of a batch split into contiguous blocks, weights replicated.

The reference is single-GPU only (train.py:111).  Rays are independent, so the path shards without any data-path
collective; exact equality with the single-GPU objective needs two exchanges per step:

  1. before backward, one all-reduce(sum) of 4 floats {sum relax*(|g|-1)^2, sum relax, sum delta_relight(*mask), count}
     because the eikonal term is a ratio of global sums (Color_NeuS.py:122-123) and the relight term the square of a
     global mean (NeuS_Trainer.py:153);
  2. after backward, ONE flat-bucket all-reduce(sum) of all parameter gradients (3.83 MiB for Color_NeuS) -- RCCL over
     xGMI on MI355X (backend "nccl"), gloo in the CPU tests -- placed before the per-parameter clip (train.py:72-73).
"""
import torch
import torch.distributed as dist


def shard_slice(n_rays: int, rank: int, world: int) -> slice:
    """Contiguous block of rays for this rank (n_rays must divide evenly so every rank does equal work)."""
    if n_rays % world != 0:
        raise ValueError(f"n_rays={n_rays} is not divisible by world size {world}")
    per = n_rays // world
    return slice(rank * per, (rank + 1) * per)


def draw_jitter(n_rays_global: int, rank: int, world: int, device):
    """Every rank draws the identical torch.rand([R_global,1]) (same CPU seed) and keeps its rows: the sharded run then
    uses exactly the jitter the single-GPU run would (NeuS.py:325)."""
    t = torch.rand([n_rays_global, 1])
    return t[shard_slice(n_rays_global, rank, world)].to(device)


def sharded_loss(out, rgb_gt, mask, n_rays_global, n_samples, group=None, lambda_fine=1.0, lambda_eikonal=0.1, lambda_mask=0.1,
                 lambda_relight=1.0, include_mask=True, eik_sums=None):
    """Loss of this rank's shard such that the SUM over ranks equals compute_loss on the whole batch, with the
    non-separable terms built from all-reduced statistics.  Returns (local_loss_for_backward, global_loss_value).

    ``eik_sums`` = (sum relax*err, sum relax) of the local shard; recovered from gradient_error when not given."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    R_loc = out["color_fine"].shape[0]
    # ---- separable terms, normalised by the GLOBAL counts
    rgb = ((out["color_fine"] - rgb_gt) ** 2).sum() / (n_rays_global * 3)
    loss = lambda_fine * rgb
    if lambda_mask != 0 and mask is not None:
        ws = out["weight_sum"].squeeze(-1).clip(1e-3, 1.0 - 1e-3)
        bce = -(mask * torch.log(ws) + (1 - mask) * torch.log(1 - ws)).sum() / n_rays_global
        loss = loss + lambda_mask * bce
    # ---- non-separable terms
    stats = torch.zeros(4, dtype=torch.float32, device=out["color_fine"].device)
    gerr = out["gradient_error"]
    if eik_sums is None:
        relax_cnt = out.get("relax_count")
        if relax_cnt is None:
            raise ValueError("sharded_loss needs out['relax_count'] (sum of the relaxed inside-sphere mask of the shard)")
        num_loc = gerr * (relax_cnt + 1e-5)
        den_loc = relax_cnt
    else:
        num_loc, den_loc = eik_sums
    stats[0], stats[1] = num_loc.detach(), den_loc.detach() if torch.is_tensor(den_loc) else den_loc
    has_rel = lambda_relight != 0 and "delta_relight" in out
    if has_rel:
        dr = out["delta_relight"]
        if include_mask and mask is not None:
            dr = dr * mask[:, None, None]
        dr_sum = dr.sum()
        stats[2] = dr_sum.detach()
    stats[3] = float(R_loc)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    den_g = stats[1] + 1e-5
    # d/d(local) of [sum_all num / (sum_all den + 1e-5)]: only the local numerator carries gradient (the mask is detached)
    eik_local = num_loc / den_g
    loss = loss + lambda_eikonal * eik_local
    glob = lambda_eikonal * stats[0] / den_g
    if has_rel:
        n_el = float(n_rays_global * n_samples * 3)
        mean_g = stats[2] / n_el
        # (mean_g)^2 with gradient only through the local part of the mean: 2*mean_g * dr_sum/n_el, written as a surrogate
        loss = loss + lambda_relight * (2.0 * mean_g * dr_sum / n_el - (2.0 * mean_g * dr_sum.detach() / n_el) + (mean_g ** 2) / world)
        glob = glob + lambda_relight * mean_g ** 2
    sep = loss.detach() - lambda_eikonal * eik_local.detach() - (lambda_relight * (mean_g ** 2) / world if has_rel else 0.0)
    sep_t = sep.clone()
    if world > 1:
        dist.all_reduce(sep_t, op=dist.ReduceOp.SUM, group=group)
    return loss, (sep_t + glob)


def allreduce_gradients(params, group=None):
    """One flat-bucket all-reduce(sum) over all parameter gradients.  The local losses are already
    normalised by global counts, so SUM (not mean) reproduces the single-GPU gradient."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
