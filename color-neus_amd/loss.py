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
