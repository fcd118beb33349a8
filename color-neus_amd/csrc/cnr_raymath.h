// Per-sample math shared by the HIP per-ray kernels and their CPU-emulation twins.
// Formulas follow the reference (file:line cited per function); derivatives are hand-derived and
// checked against autograd of the oracle in tests/.
#pragma once
#include "cnr_common.h"

namespace cnr {

// torch.linspace(start, end, steps)[i] in float32 (symmetric evaluation used by ATen's CPU/GPU kernels)
CNR_HD float linspace_at(float start, float end, int steps, int i) {
  if (steps <= 1) return start;
  float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// positional encoding row: [x, sin(2^0 x), cos(2^0 x), ...]                (PositionEncoding.py:51-76)
CNR_HD void pe_row(const float x[3], int multires, float* out /* 3 + 6*multires */) {
  out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
  float f = 1.0f;
  for (int k = 0; k < multires; ++k) {
    for (int c = 0; c < 3; ++c) {
      float a = x[c] * f, sn, cs;
      sincosf(a, &sn, &cs);   // (one argument reduction for the pair)
      out[3 + 6 * k + c] = sn;
      out[6 + 6 * k + c] = cs;
    }
    f *= 2.0f;
  }
}

// S-density alpha of one section                                  (Color_NeuS.py:69-90 / NeuS.py:236-256)
struct AlphaOut {
  float tc, ic, pe, ne, pc, nc, a_raw, alpha;
};
CNR_HD AlphaOut alpha_forward(float sdf, const float g[3], const float d[3], float dist, float inv_s, float r) {
  AlphaOut o;
  o.tc = d[0] * g[0] + d[1] * g[1] + d[2] * g[2];
  float A = fmaxf(-o.tc * 0.5f + 0.5f, 0.0f);
  float B = fmaxf(-o.tc, 0.0f);
  o.ic = -(A * (1.0f - r) + B * r);
  o.ne = sdf + o.ic * dist * 0.5f;
  o.pe = sdf - o.ic * dist * 0.5f;
  o.pc = sigmoidf_(o.pe * inv_s);
  o.nc = sigmoidf_(o.ne * inv_s);
  o.a_raw = (o.pc - o.nc + 1e-5f) / (o.pc + 1e-5f);
  o.alpha = fminf(fmaxf(o.a_raw, 0.0f), 1.0f);
  return o;
}

// reverse of alpha_forward: given d alpha and d prev_cdf (cdf_fine upstream) produce d sdf, d tc, d inv_s, d dist
struct AlphaGrad {
  float d_sdf, d_tc, d_inv_s, d_dist;
};
CNR_HD AlphaGrad alpha_backward(const AlphaOut& o, float dist, float inv_s, float r, float d_alpha, float d_pc_extra) {
  AlphaGrad gr;
  float da = (o.a_raw >= 0.0f && o.a_raw <= 1.0f) ? d_alpha : 0.0f;
  float den = o.pc + 1e-5f;
  float d_pc = da * (o.nc / (den * den)) + d_pc_extra;
  float d_nc = -da / den;
  float spc = o.pc * (1.0f - o.pc), snc = o.nc * (1.0f - o.nc);
  float d_pe = d_pc * spc * inv_s;
  float d_ne = d_nc * snc * inv_s;
  gr.d_inv_s = d_pc * spc * o.pe + d_nc * snc * o.ne;
  gr.d_sdf = d_pe + d_ne;
  float d_ic = (d_ne - d_pe) * dist * 0.5f;
  gr.d_dist = (d_ne - d_pe) * o.ic * 0.5f;
  float dic_dtc = (o.tc < 1.0f ? 0.5f * (1.0f - r) : 0.0f) + (o.tc < 0.0f ? r : 0.0f);
  gr.d_tc = d_ic * dic_dtc;
  return gr;
}

// The same pair in double, for the per-ray background compositor (N_OUTSIDE > 0: one thread per ray, speed is no concern there): the
// d inv_s terms of one ray cancel to a few percent of their size, so their float32 round-off would show in d variance.
struct AlphaOutD { double tc, ic, pe, ne, pc, nc, a_raw; };
CNR_HD AlphaOutD alpha_forward_d(double sdf, const float g[3], const float d[3], double dist, double inv_s, double r) {
  AlphaOutD o;
  o.tc = (double)d[0] * g[0] + (double)d[1] * g[1] + (double)d[2] * g[2];
  const double A = fmax(-o.tc * 0.5 + 0.5, 0.0), B = fmax(-o.tc, 0.0);
  o.ic = -(A * (1.0 - r) + B * r);
  o.ne = sdf + o.ic * dist * 0.5;
  o.pe = sdf - o.ic * dist * 0.5;
  o.pc = 1.0 / (1.0 + exp(-o.pe * inv_s));
  o.nc = 1.0 / (1.0 + exp(-o.ne * inv_s));
  o.a_raw = (o.pc - o.nc + 1e-5) / (o.pc + 1e-5);
  return o;
}
CNR_HD AlphaGrad alpha_backward_d(const AlphaOutD& o, double dist, double inv_s, double r, double d_alpha, double d_pc_extra, double* d_inv_s) {
  AlphaGrad gr;
  const double da = (o.a_raw >= 0.0 && o.a_raw <= 1.0) ? d_alpha : 0.0;
  const double den = o.pc + 1e-5;
  const double d_pc = da * (o.nc / (den * den)) + d_pc_extra, d_nc = -da / den;
  const double spc = o.pc * (1.0 - o.pc), snc = o.nc * (1.0 - o.nc);
  const double d_pe = d_pc * spc * inv_s, d_ne = d_nc * snc * inv_s;
  *d_inv_s = d_pc * spc * o.pe + d_nc * snc * o.ne;
  gr.d_inv_s = (float)*d_inv_s;
  gr.d_sdf = (float)(d_pe + d_ne);
  const double d_ic = (d_ne - d_pe) * dist * 0.5;
  gr.d_dist = (float)((d_ne - d_pe) * o.ic * 0.5);
  const double dic_dtc = (o.tc < 1.0 ? 0.5 * (1.0 - r) : 0.0) + (o.tc < 0.0 ? r : 0.0);
  gr.d_tc = (float)(d_ic * dic_dtc);
  return gr;
}

// up-sampling section alpha (no clip)                                              (NeuS.py:144-177)
CNR_HD float upsample_alpha(float s0, float s1, float z0, float z1, float cos_val, float inv_s) {
  float mid = (s0 + s1) * 0.5f;
  float dist = z1 - z0;
  float pe = mid - cos_val * dist * 0.5f;
  float ne = mid + cos_val * dist * 0.5f;
  float pc = sigmoidf_(pe * inv_s), nc = sigmoidf_(ne * inv_s);
  return (pc - nc + 1e-5f) / (pc + 1e-5f);
}

// One ray of get_rays_multicam / get_rays_at (ray_utils.py:16-119): camera-frame direction of pixel (px, py), optional normalisation,
// rotation into the world frame, camera centre as origin.  u = unnormalised direction, dirs = (normalised) camera-frame direction.
struct RayGeom { float u[3], un, dirs[3], d[3], o[3]; };
CNR_HD RayGeom ray_geometry(const float* c2w /* one [4][4] */, float fx, float fy, int H, int W, int px, int py, int normalize, int opengl) {
  RayGeom r;
  const float ys = opengl ? -1.0f : 1.0f, zs = opengl ? -1.0f : 1.0f;
  r.u[0] = ((float)px - (float)W * 0.5f) / fx;
  r.u[1] = ys * ((float)py - (float)H * 0.5f) / fy;
  r.u[2] = zs;
  r.un = sqrtf(r.u[0] * r.u[0] + r.u[1] * r.u[1] + r.u[2] * r.u[2]);
  for (int k = 0; k < 3; ++k) r.dirs[k] = normalize ? r.u[k] / r.un : r.u[k];
  for (int k = 0; k < 3; ++k) {
    r.d[k] = r.dirs[0] * c2w[k * 4] + r.dirs[1] * c2w[k * 4 + 1] + r.dirs[2] * c2w[k * 4 + 2];
    r.o[k] = c2w[k * 4 + 3];
  }
  return r;
}

// d/d rgb of inverse_sigmoid (clamp semantics of torch: gradient passes where the clamp is inactive, bounds inclusive)
CNR_HD float inverse_sigmoid_grad(float rgb) {
  if (rgb < 0.0f || rgb > 1.0f) return 0.0f;
  float x = rgb;
  float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.0f - x, 1e-5f);
  float g1 = x >= 1e-5f ? 1.0f / x1 : 0.0f;
  float g2 = (1.0f - x) >= 1e-5f ? 1.0f / x2 : 0.0f;
  return g1 + g2;
}

}  // namespace cnr
