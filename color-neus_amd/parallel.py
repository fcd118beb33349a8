"""Ray-sharded data parallelism: one process per GPU, rays of a batch split into contiguous blocks, weights replicated.

The reference is single-GPU only (train.py:111).  Rays are independent, so the path shards without any data-path
collective; exact equality with the single-GPU objective needs two exchanges per step:

  1. before backward, one small all-reduce(sum): 3 floats {sum relax*(|g|-1)^2, sum relax, sum delta_relight(*mask)} in
     sharded_loss below (torch loss; the separable terms are normalised locally), 5 floats in loss.compute_loss_fused
     (the three cnr_loss_sums + the two eikonal sums) -- needed because the eikonal term is a ratio of global sums
     (Color_NeuS.py:122-123) and the relight term the square of a global mean (NeuS_Trainer.py:153);
  2. after backward, ONE flat-bucket all-reduce(sum) of all parameter gradients (3.83 MiB for Color_NeuS) -- RCCL over
     xGMI on MI355X (backend "nccl"), gloo in the CPU tests -- placed before the per-parameter clip (train.py:72-73).
"""
import torch
import torch.distributed as dist


def _world(group=None):
    return (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)


def _gather_rows(local, counts, dst, group):
    """Blocks of rows of unequal height from every rank -> one tensor on rank dst (None elsewhere): padded to the tallest block, gathered, trimmed."""
    rank, world = _world(group)
    if world == 1:
        return local
    hmax = max(counts)
    pad = torch.zeros((hmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def sharded_extract_fields(renderer, bound_min, bound_max, device, resolution, group=None, dst=0):
    """extract_fields (NeuS.py:14-28) over the ranks of a node: rank r evaluates a slab of lattice rows x, the slabs are gathered on rank
    ``dst`` (returns the [res, res, res] volume there, None elsewhere).  No data-path collective besides that gather."""
    rank, world = _world(group)
    per = (resolution + world - 1) // world
    bounds = [(min(r * per, resolution), min((r + 1) * per, resolution)) for r in range(world)]
    x0, x1 = bounds[rank]
    if x1 > x0:
        slab = renderer.extract_fields_slab(bound_min, bound_max, device, resolution, x0, x1)
    else:
        slab = torch.empty(0, resolution, resolution, dtype=torch.float32, device=torch.device(device))
    return _gather_rows(slab, [b - a for a, b in bounds], dst, group)


def sharded_render_image(renderer, rays_o, rays_d, near, far, chunk=1024, group=None, dst=0, keys=("color_fine", "depth"), **render_kw):
    """validate_image's render loop (NeuS_Trainer.py:233-245: all rays of a view in EVAL_RAY_SIZE chunks, only colour and depth consumed)
    over the ranks of a node: contiguous, chunk-aligned blocks of chunks per rank, results gathered on rank ``dst`` ({key: tensor} there,
    None elsewhere).  Every rank holds the full ray set (cheap: 32 B / ray) and walks ALL chunks in order so that the per-chunk jitter
    draws from the CPU generator (the reference jitters in eval mode too, NeuS.py:320-326) are the ones a single process makes; it renders
    only its own chunks."""
    rank, world = _world(group)
    n = rays_o.shape[0]
    # row widths of the keys a rank may gather: a rank that owns no chunk must still present a buffer of the right shape to the collective
    m = int(renderer.rcfg.n_total) if hasattr(renderer, "rcfg") else None
    widths = {"color_fine": 3, "global_color": 3, "depth": 1, "weight_sum": 1, "weight_max": 1, "s_val": 1}
    if m is not None:
        # (with N_OUTSIDE > 0 the weights cover the background samples too: NeuS.py:262-268)
        widths.update({"weights": m + int(getattr(renderer, "n_outside", 0) or 0), "cdf_fine": m, "inside_sphere": m, "z_vals": m, "gradients": 3 * m, "delta_relight": 3 * m})
    unknown = [k for k in keys if k not in widths]
    if unknown:
        raise ValueError(f"sharded_render_image: no per-ray width known for keys {unknown} (gatherable: {sorted(widths)})")
    nchunks = (n + chunk - 1) // chunk
    per = (nchunks + world - 1) // world
    c0, c1 = min(rank * per, nchunks), min((rank + 1) * per, nchunks)
    perturb = render_kw.get("perturb_overwrite", -1)
    draws = perturb != 0 and (perturb > 0 or renderer.perturb > 0)
    parts = {k: [] for k in keys}
    with torch.no_grad():
        for c in range(nchunks):
            a, b = c * chunk, min((c + 1) * chunk, n)
            if c0 <= c < c1:
                out = renderer(rays_o[a:b], rays_d[a:b], near[a:b], far[a:b], **render_kw)
                for k in keys:
                    parts[k].append(out[k].reshape(b - a, -1))
            elif draws:   # another rank's chunk: keep the generator in step (NeuS.py:325, and :335 for the background samples)
                torch.rand([b - a, 1])
                if getattr(renderer, "n_outside", 0) > 0:
                    torch.rand([b - a, renderer.n_outside])
    counts = [max(0, min((r + 1) * per, nchunks) * chunk - min(r * per, nchunks) * chunk) for r in range(world)]
    counts = [min(cnt, max(0, n - min(r * per, nchunks) * chunk)) for r, cnt in enumerate(counts)]
    res = {}
    for k in keys:
        local = torch.cat(parts[k], 0) if parts[k] else torch.empty(0, widths[k], dtype=torch.float32, device=rays_o.device)
        if local.shape[1] != widths[k]:
            raise RuntimeError(f"sharded_render_image: key {k} has {local.shape[1]} columns per ray, expected {widths[k]}")
        res[k] = _gather_rows(local, counts, dst, group)
    return res if rank == dst else None


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
                 lambda_relight=1.0, include_mask=True):
    """Loss of this rank's ray shard, built so that the SUM over ranks of its gradients equals the gradient of
    compute_loss (loss.py) on the whole batch.  Returns (local_loss_for_backward, global_loss_value_detached).

    Needs out["eik_sums"] = {sum relax*(|g|-1)^2, sum relax} of the shard (extra output of the native renderer)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = out["color_fine"].device
    # ---- separable terms, normalised by GLOBAL counts
    local = lambda_fine * ((out["color_fine"] - rgb_gt) ** 2).sum() / (n_rays_global * 3)
    if lambda_mask != 0 and mask is not None:
        ws = out["weight_sum"].squeeze(-1).clip(1e-3, 1.0 - 1e-3)
        local = local + lambda_mask * (-(mask * torch.log(ws) + (1 - mask) * torch.log(1 - ws)).sum() / n_rays_global)
    # ---- statistics of the non-separable terms: one 3-float all-reduce before backward (the fused loss sends 5, see loss.py)
    has_rel = lambda_relight != 0 and "delta_relight" in out
    dr_sum = None
    stats = torch.zeros(3, dtype=torch.float32, device=dev)
    stats[0:2] = out["eik_sums"].detach()
    if has_rel:
        dr = out["delta_relight"]
        if include_mask and mask is not None:
            dr = dr * mask[:, None, None]
        dr_sum = dr.sum()
        stats[2] = dr_sum.detach()
    den_loc = stats[1].clone()
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    # eikonal: global ratio sum_r num_r / (sum_r den_r + 1e-5); the local output is num_r / (den_r + 1e-5)
    eik_local = out["gradient_error"] * ((den_loc + 1e-5) / (stats[1] + 1e-5))
    local = local + lambda_eikonal * eik_local
    value = local.detach().clone()
    if world > 1:
        dist.all_reduce(value, op=dist.ReduceOp.SUM, group=group)
    if has_rel:
        n_el = float(n_rays_global * n_samples * 3)
        mean_g = stats[2] / n_el
        # d/d(shard) of mean_g^2 = 2 * mean_g * d(dr_sum_local)/n_el : linear surrogate with the right gradient
        local = local + lambda_relight * 2.0 * mean_g * (dr_sum - dr_sum.detach()) / n_el
        value = value + lambda_relight * mean_g ** 2
    return local, value


def allreduce_gradients(params, group=None):
    """One flat-bucket all-reduce(sum) over all parameter gradients.  The local losses are already
    normalised by global counts, so SUM (not mean) reproduces the single-GPU gradient.

    The renderer's backward lays the gradients out in one flat buffer (renderer._RenderFunction.backward), so the collective runs on
    that buffer in place; gradients that do not tile one buffer (a model with extra parameters) fall back to a gathered copy."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    from .optim import flat_view_of_grads
    with_grad = [p for p in params if p.grad is not None]
    flat = flat_view_of_grads(with_grad)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return
    grads = [p.grad for p in with_grad]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
