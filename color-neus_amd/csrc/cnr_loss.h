// Per-element pieces of the training loss (NeuS_Trainer.compute_loss, NeuS_Trainer.py:129-171), shared by the HIP kernels and the CPU
// emulation used in the tests.
#pragma once
#include "cnr_backend.h"

namespace cnr {

CNR_HD float loss_rgb_term(float c, float gt, int l1) { const float e = c - gt; return l1 ? fabsf(e) : e * e; }
CNR_HD float loss_rgb_grad(float c, float gt, int l1) { const float e = c - gt; return l1 ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : e; }
// F.binary_cross_entropy(clip(ws, 1e-3, 1 - 1e-3), m): torch clamps the logs at -100, irrelevant inside the clip range
CNR_HD float loss_bce_term(float ws, float m) {
  const float w = fminf(fmaxf(ws, 1e-3f), 1.0f - 1e-3f);
  return -(m * logf(w) + (1.0f - m) * logf(1.0f - w));
}
CNR_HD float loss_bce_grad(float ws, float m) {   // clip passes the gradient only strictly inside... torch.clamp passes it on the closed range
  if (ws < 1e-3f || ws > 1.0f - 1e-3f) return 0.0f;
  return -m / ws + (1.0f - m) / (1.0f - ws);
}

// the scalar arithmetic of compute_loss around the two reduction phases (fp32, operation order of NeuS_Trainer.py:146-171)
CNR_HD void loss_combine(const LossScalars& c, const float* sums, const float* gerr, float* out) {
  const float rgb = sums[0] / c.den_rgb;
  const float eik = gerr[0];
  float loss = c.lf * rgb + c.le * eik;
  float mask = 0.0f, rel = 0.0f, mean_rel = 0.0f;
  if (c.use_mask) { mask = sums[1] / c.Rg; loss = loss + c.lm * mask; }
  if (c.use_relight) { mean_rel = sums[2] / c.den_rel; rel = mean_rel * mean_rel; loss = loss + c.lr * rel; }
  out[0] = loss; out[1] = rgb; out[2] = eik; out[3] = mask; out[4] = rel; out[5] = mean_rel;
}
// ray-sharded runs: the scalar tail on the all-reduced statistics (stats[0..4] summed over the ranks, stats[5] = this rank's own sum of the relax mask).
// The eikonal term is the ratio of the GLOBAL sums (Color_NeuS.py:122-123); out[6] = d (global ratio) / d (this rank's gradient_error output)
// = (den_local + 1e-5) / (den_global + 1e-5), because the rank's output is num_local / (den_local + 1e-5).
CNR_HD void loss_shard_combine(const LossScalars& c, const float* stats, float* out) {
  const float den = stats[4] + 1e-5f;
  const float eik = stats[3] / den;
  loss_combine(c, stats, &eik, out);
  out[6] = (stats[5] + 1e-5f) / den;
  out[7] = 0.0f;
}
CNR_HD void loss_coef(const LossScalars& c, const float* g_loss, const float* mean_rel, const float* eik_factor /* or null */, float* coef) {
  const float g = g_loss[0];
  coef[0] = g * c.c_rgb;
  coef[1] = c.use_mask ? g * c.c_bce : 0.0f;
  coef[2] = c.use_relight ? g * c.c_rel * mean_rel[0] : 0.0f;
  coef[3] = eik_factor ? g * c.le * eik_factor[0] : g * c.le;
}

}  // namespace cnr
