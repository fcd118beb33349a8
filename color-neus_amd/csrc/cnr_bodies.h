// Per-element bodies of the point-wise kernels (host+device; shared by the HIP and CPU-emulation backends).
#pragma once
#include "cnr_backend.h"
#include "cnr_raymath.h"

namespace cnr {

// one (ray, sample) element of EmbedZ; `out`: where the kEmb-wide row goes (the HIP kernels stage rows in LDS and store them coalesced)
CNR_HD void body_embed_z_rows(const EmbedZ& p, long idx, float* out, float*) {
  long r = idx / p.m;
  int j = (int)(idx - r * p.m);
  float zz;
  if (p.make_z) {
    float nr = p.near_[r], fr = p.far_[r];
    zz = nr + (fr - nr) * linspace_at(0.0f, 1.0f, p.m, j);
    if (p.t_rand) zz = zz + (p.t_rand[r] - 0.5f) * 2.0f / (float)p.m;
    p.z[r * p.ldz + j] = zz;
  } else {
    zz = p.z[r * p.ldz + j];
  }
  float x[3];
  for (int c = 0; c < 3; ++c) x[c] = (p.o[r * 3 + c] + p.d[r * 3 + c] * zz) * p.scale;
  for (int c = 0; c < kEmb; ++c) out[c] = 0.0f;
  pe_row(x, p.multires, out);   // (straight into the destination row: a local array indexed by the runtime multires would live in scratch memory)
}
CNR_HD void body_embed_z(const EmbedZ& p, long idx) { body_embed_z_rows(p, idx, p.E + idx * kEmb, nullptr); }

CNR_HD void body_embed_pts_rows(const EmbedPts& p, long i, float* e, float* a) {
  float pp[3];
  if (p.pts) {
    pp[0] = p.pts[i * 3]; pp[1] = p.pts[i * 3 + 1]; pp[2] = p.pts[i * 3 + 2];
  } else {
    long idx = p.start + i;
    long r2 = (long)p.res * p.res;
    int ix = (int)(idx / r2), iy = (int)((idx / p.res) % p.res), iz = (int)(idx % p.res);
    pp[0] = linspace_at(p.bmin[0], p.bmax[0], p.res, ix);
    pp[1] = linspace_at(p.bmin[1], p.bmax[1], p.res, iy);
    pp[2] = linspace_at(p.bmin[2], p.bmax[2], p.res, iz);
  }
  float x[3] = {pp[0] * p.scale, pp[1] * p.scale, pp[2] * p.scale};
  for (int c = 0; c < kEmb; ++c) e[c] = 0.0f;
  pe_row(x, p.multires, e);
  if (p.AUX) {
    for (int c = 0; c < kAux; ++c) a[c] = 0.0f;
    a[0] = pp[0]; a[1] = pp[1]; a[2] = pp[2];
  }
}
CNR_HD void body_embed_pts(const EmbedPts& p, long i) { body_embed_pts_rows(p, i, p.E + i * kEmb, p.AUX ? p.AUX + i * kAux : nullptr); }

CNR_HD void body_fine_setup_rows(const FineSetup& p, long pt, float* e, float* a) {
  long r = pt / p.M;
  int j = (int)(pt - r * p.M);
  float z0 = p.z[r * p.M + j];
  float dist = (j + 1 < p.M) ? p.z[r * p.M + j + 1] - z0 : p.sample_dist;
  float mid = z0 + dist * 0.5f;
  float pp[3], x[3], dd[3];
  for (int c = 0; c < 3; ++c) {
    dd[c] = p.d[r * 3 + c];
    pp[c] = p.o[r * 3 + c] + dd[c] * mid;
    x[c] = pp[c] * p.scale;
  }
  for (int c = 0; c < kEmb; ++c) e[c] = 0.0f;
  pe_row(x, p.multires, e);
  for (int c = 0; c < kAux; ++c) a[c] = 0.0f;
  a[0] = pp[0]; a[1] = pp[1]; a[2] = pp[2];
  if (p.multires_view > 0) pe_row(dd, p.multires_view, a + 6);
  else { a[6] = dd[0]; a[7] = dd[1]; a[8] = dd[2]; }
}
CNR_HD void body_fine_setup(const FineSetup& p, long pt) { body_fine_setup_rows(p, pt, p.E + pt * kEmb, p.AUX + pt * kAux); }

// g = scale * J^T ce with J = d PE / d x0; E holds sin/cos of the encoding
CNR_HD void body_grad_finish(const GradFinish& p, long pt) {
  const float* e = p.E + pt * kEmb;
  const float* c0 = p.ce0 + pt * kEmb;
  const float* cs = p.ces ? p.ces + pt * kEmb : nullptr;
  for (int c = 0; c < 3; ++c) {
    float acc = c0[c] + (cs ? cs[c] : 0.0f);
    float f = 1.0f;
    for (int k = 0; k < p.multires; ++k) {
      float csin = c0[3 + 6 * k + c] + (cs ? cs[3 + 6 * k + c] : 0.0f);
      float ccos = c0[6 + 6 * k + c] + (cs ? cs[6 + 6 * k + c] : 0.0f);
      acc = acc + f * (e[6 + 6 * k + c] * csin - e[3 + 6 * k + c] * ccos);
      f *= 2.0f;
    }
    float g = acc * p.scale;
    p.grad_out[pt * 3 + c] = g;
    p.AUX[pt * kAux + 3 + c] = g;
  }
  if (p.neg_g_as_view) {
    float v[3] = {-p.AUX[pt * kAux + 3], -p.AUX[pt * kAux + 4], -p.AUX[pt * kAux + 5]};
    float row[kAux];
    for (int c = 0; c < kAux; ++c) row[c] = 0.0f;
    if (p.multires_view > 0) pe_row(v, p.multires_view, row);
    else { row[0] = v[0]; row[1] = v[1]; row[2] = v[2]; }
    int npe = p.multires_view > 0 ? 3 + 6 * p.multires_view : 3;
    for (int c = 0; c < npe; ++c) p.AUX[pt * kAux + 6 + c] = row[c];
  }
  if (p.featx) {
    float* fx = p.featx + pt * p.ldfx + p.F;
    const float* a = p.AUX + pt * kAux;
    const int w = p.ldfx - p.F;
    for (int c = 0; c < w; ++c) fx[c] = c < kAux ? a[c] : 0.0f;
  }
}

CNR_HD void body_coltop_bwd(const ColTopBwd& p, long pt) {
  for (int c = 0; c < 3; ++c) {
    float gc = p.gc_a[pt * kTop + c] + (p.gc_b ? p.gc_b[pt * 4 + c] : 0.0f);
    float y = p.gcolor[pt * 4 + c];
    p.out[pt * kTop + c] = p.squeeze ? gc * y * (1.0f - y) : gc;
  }
  for (int c = 3; c < kTop; ++c) p.out[pt * kTop + c] = 0.0f;
}

CNR_HD void body_gbar_finish(const GbarFinish& p, long pt) {
  const float* e = p.E + pt * kEmb;
  float* cb = p.cbar + pt * kEmb;
  for (int c = 0; c < kEmb; ++c) cb[c] = 0.0f;
  for (int c = 0; c < 3; ++c) {
    float gb = p.gbar_alpha[pt * 4 + c];
    if (p.daux_c) gb += p.daux_c[pt * kAux + 3 + c];
    if (p.daux_r) gb += p.daux_r[pt * kAux + 3 + c];
    p.gbar_total[pt * 4 + c] = gb;
    float t = gb * p.scale;
    cb[c] = t;
    float f = 1.0f;
    for (int k = 0; k < p.multires; ++k) {
      cb[3 + 6 * k + c] = f * e[6 + 6 * k + c] * t;     // d g / d ce_sin = 2^k cos(2^k x0)
      cb[6 + 6 * k + c] = -f * e[3 + 6 * k + c] * t;    // d g / d ce_cos = -2^k sin(2^k x0)
      f *= 2.0f;
    }
  }
  p.gbar_total[pt * 4 + 3] = 0.0f;
}

// total cotangent of the sample position p (only evaluated when rays require grad)
CNR_HD void body_pbar_finish(const PbarFinish& p, long pt) {
  const float* e = p.E + pt * kEmb;
  for (int c = 0; c < 3; ++c) {
    float eb = p.ebar0[pt * kEmb + c] + (p.ebars ? p.ebars[pt * kEmb + c] : 0.0f);
    float x0bar = eb;
    float second = 0.0f;
    float f = 1.0f;
    for (int k = 0; k < p.multires; ++k) {
      float s = e[3 + 6 * k + c], co = e[6 + 6 * k + c];
      float ebs = p.ebar0[pt * kEmb + 3 + 6 * k + c] + (p.ebars ? p.ebars[pt * kEmb + 3 + 6 * k + c] : 0.0f);
      float ebc = p.ebar0[pt * kEmb + 6 + 6 * k + c] + (p.ebars ? p.ebars[pt * kEmb + 6 + 6 * k + c] : 0.0f);
      x0bar += f * (co * ebs - s * ebc);
      float ces = p.ce0[pt * kEmb + 3 + 6 * k + c] + (p.ces ? p.ces[pt * kEmb + 3 + 6 * k + c] : 0.0f);
      float cec = p.ce0[pt * kEmb + 6 + 6 * k + c] + (p.ces ? p.ces[pt * kEmb + 6 + 6 * k + c] : 0.0f);
      second += f * f * (-s * ces - co * cec);
      f *= 2.0f;
    }
    x0bar += p.gbar_total[pt * 4 + c] * p.scale * second;
    float pb = x0bar * p.scale;
    if (p.daux_c) pb += p.daux_c[pt * kAux + c];
    if (p.daux_r) pb += p.daux_r[pt * kAux + c];
    p.pbar[pt * 4 + c] = pb;
  }
  p.pbar[pt * 4 + 3] = 0.0f;
}

// one ray of GenRays
CNR_HD void body_gen_rays(const GenRays& p, long i) {
  const long hw = (long)p.H * p.W;
  const long idx = p.pix_idx ? p.pix_idx[i] : i;
  const int cam = (int)(idx / hw);
  const long pix = idx - (long)cam * hw;
  const int py = (int)(pix / p.W), px = (int)(pix - (long)py * p.W);
  const RayGeom r = ray_geometry(p.c2w + (long)cam * 16, p.focal[0], p.focal[1], p.H, p.W, px, py, p.normalize, p.opengl);
  float o[3];
  for (int k = 0; k < 3; ++k) {
    o[k] = p.origin ? (r.o[k] - p.origin[k]) / p.radius : r.o[k];
    p.rays_o[i * 3 + k] = o[k];
    p.rays_d[i * 3 + k] = r.d[k];
  }
  if (p.rgb) for (int k = 0; k < 3; ++k) p.rgb[i * 3 + k] = p.image[idx * 3 + k];
  if (p.mask_sel) p.mask_sel[i] = p.mask[idx];
  if (p.near_ && p.far_) {   // near_far_from_sphere (ray_utils.py:7-13)
    const float a = r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2];
    const float b = 2.0f * (o[0] * r.d[0] + o[1] * r.d[1] + o[2] * r.d[2]);
    const float mid = 0.5f * (-b) / a;
    p.near_[i] = mid - 1.0f;
    p.far_[i] = mid + 1.0f;
  }
}

// gradient contributions of one ray: out[0..11] = d c2w[cam][k][0..3] (k = 0..2), out[12..13] = d focal
CNR_HD void body_gen_rays_bwd1(const GenRaysBwd& q, long i, int* cam_out, float out[14]) {
  const GenRays& p = q.f;
  const long hw = (long)p.H * p.W;
  const long idx = p.pix_idx ? p.pix_idx[i] : i;
  const int cam = (int)(idx / hw);
  const long pix = idx - (long)cam * hw;
  const int py = (int)(pix / p.W), px = (int)(pix - (long)py * p.W);
  const float* c2w = p.c2w + (long)cam * 16;
  const float fx = p.focal[0], fy = p.focal[1];
  const RayGeom r = ray_geometry(c2w, fx, fy, p.H, p.W, px, py, p.normalize, p.opengl);
  float dO[3], dD[3];
  for (int k = 0; k < 3; ++k) { dO[k] = q.d_rays_o ? q.d_rays_o[i * 3 + k] : 0.0f; dD[k] = q.d_rays_d ? q.d_rays_d[i * 3 + k] : 0.0f; }
  if (q.d_near && q.d_far) {   // mid = -(o.d) / (d.d): d mid / d o = -d / a, d mid / d d = (-o - 2 mid d) / a
    float o[3];
    for (int k = 0; k < 3; ++k) o[k] = p.origin ? (r.o[k] - p.origin[k]) / p.radius : r.o[k];
    const float a = r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2];
    const float mid = -(o[0] * r.d[0] + o[1] * r.d[1] + o[2] * r.d[2]) / a;
    const float dm = q.d_near[i] + q.d_far[i];
    for (int k = 0; k < 3; ++k) { dO[k] += dm * (-r.d[k] / a); dD[k] += dm * (-o[k] - 2.0f * mid * r.d[k]) / a; }
  }
  float ddirs[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < 3; ++k) {
    for (int j = 0; j < 3; ++j) { out[k * 4 + j] = dD[k] * r.dirs[j]; ddirs[j] += dD[k] * c2w[k * 4 + j]; }
    out[k * 4 + 3] = p.origin ? dO[k] / p.radius : dO[k];
  }
  float du[3];
  if (p.normalize) {
    const float dot = r.dirs[0] * ddirs[0] + r.dirs[1] * ddirs[1] + r.dirs[2] * ddirs[2];
    for (int j = 0; j < 3; ++j) du[j] = (ddirs[j] - r.dirs[j] * dot) / r.un;
  } else {
    for (int j = 0; j < 3; ++j) du[j] = ddirs[j];
  }
  out[12] = -r.u[0] / fx * du[0];
  out[13] = -r.u[1] / fy * du[1];
  *cam_out = cam;
}

}  // namespace cnr
