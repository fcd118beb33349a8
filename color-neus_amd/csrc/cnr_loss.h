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

}  // namespace cnr
