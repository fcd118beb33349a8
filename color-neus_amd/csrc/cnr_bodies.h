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
    float gc = p.gc_a[pt * p.ldtop + c] + (p.gc_b ? p.gc_b[pt * 4 + c] : 0.0f);
    float y = p.gcolor[pt * 4 + c];
    p.out[pt * p.ldtop + c] = p.squeeze ? gc * y * (1.0f - y) : gc;
  }
  for (int c = 3; c < p.ldtop; ++c) p.out[pt * p.ldtop + c] = 0.0f;
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
// An index outside [0, n_cams * H * W) (only a caller-supplied list can hold one) reads nothing: every output of that ray is NaN -- which
// the loss then shows -- where the reference's torch indexing would raise; the backward kernel gives such a ray no contribution.
CNR_HD void body_gen_rays(const GenRays& p, long i) {
  const long hw = (long)p.H * p.W;
  const long idx = p.pix_idx ? p.pix_idx[i] : i;
  if (idx < 0 || idx >= hw * p.n_cams) {
    const float nan_ = __builtin_nanf("");
    for (int k = 0; k < 3; ++k) { p.rays_o[i * 3 + k] = nan_; p.rays_d[i * 3 + k] = nan_; if (p.rgb) p.rgb[i * 3 + k] = nan_; }
    if (p.mask_sel) p.mask_sel[i] = nan_;
    if (p.near_ && p.far_) { p.near_[i] = nan_; p.far_[i] = nan_; }
    // the reference's torch indexing raises here; the device cannot: the ray is NaN (loud in the loss) AND counted, so that the host can raise at
    // its next synchronisation point (rays.raise_if_bad_indices) before the NaN gradients reach the optimiser state
    if (p.bad_count) {
#if defined(CNR_CPU_EMU)
#pragma omp atomic
      *p.bad_count += 1;
#else
      atomicAdd(p.bad_count, 1);
#endif
    }
    return;
  }
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
  if (idx < 0 || idx >= hw * p.n_cams) {   // (see body_gen_rays: no camera owns this ray)
    for (int k = 0; k < 14; ++k) out[k] = 0.0f;
    *cam_out = -1;
    return;
  }
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

// ================================================================================================
// N_OUTSIDE > 0: the NeRF++ background (NeuS.py:95-134, 313-369).  Plain per-ray / per-point code: no shipped configuration takes this path.
// ================================================================================================

// the inverse-depth positions zz[k] in (0, 1) of the background samples before the flip (NeuS.py:316, 331-336)
CNR_HD void outside_zz(int n, const float* t /* [n] or null */, float* zz) {
  const float hi = (float)(1.0 - 1.0 / ((double)n + 1.0));
  for (int k = 0; k < n; ++k) zz[k] = linspace_at(1e-3f, hi, n, k);
  if (t) {
    float lo_[kMaxOutside], up_[kMaxOutside];
    for (int k = 0; k < n; ++k) {
      lo_[k] = k == 0 ? zz[0] : 0.5f * (zz[k] + zz[k - 1]);
      up_[k] = k == n - 1 ? zz[n - 1] : 0.5f * (zz[k + 1] + zz[k]);
    }
    for (int k = 0; k < n; ++k) zz[k] = lo_[k] + (up_[k] - lo_[k]) * t[k];
  }
}
CNR_HD void body_outside_z(const OutsideZ& p, long ray) {
  const int n = p.n_out, M = p.M;
  float zz[kMaxOutside], zo[kMaxOutside];
  outside_zz(n, p.t_rand ? p.t_rand + ray * n : nullptr, zz);
  const float fr = p.far_[ray], inv_n = 1.0f / (float)p.n_samples;
  for (int k = 0; k < n; ++k) zo[k] = fr / zz[n - 1 - k] + inv_n;          // far / flip(zz) + 1 / n_samples (NeuS.py:338)
  const float* z = p.z + ray * M;
  float* zf = p.z_feed + ray * (M + n);
  int* src = p.src + ray * (M + n);
  int a = 0, b = 0;
  for (int i = 0; i < M + n; ++i) {                                          // torch.sort of the concatenation = merge of two ascending lists
    const bool take_z = b >= n || (a < M && z[a] <= zo[b]);
    if (take_z) { zf[i] = z[a]; src[i] = a; ++a; } else { zf[i] = zo[b]; src[i] = -1 - b; ++b; }
  }
}
CNR_HD void body_outside_z_bwd(const OutsideZBwd& p, long ray) {
  const int n = p.n_out, M = p.M;
  float zz[kMaxOutside];
  outside_zz(n, p.t_rand ? p.t_rand + ray * n : nullptr, zz);
  const int* src = p.src + ray * (M + n);
  const float* dz = p.d_z_feed + ray * (M + n);
  float dfar = 0.0f;
  for (int i = 0; i < M + n; ++i) {
    if (src[i] >= 0) { if (p.d_z) p.d_z[ray * M + src[i]] = dz[i]; }
    else dfar += dz[i] / zz[n - 1 - (-1 - src[i])];
  }
  p.d_far[ray] = dfar;
}

// section geometry of one background sample (NeuS.py:102-111)
struct BgGeom { float dist, mid, pts[3], rn, r, q[4]; bool clipped; };
CNR_HD BgGeom bg_geometry(const float* o, const float* d, const float* zf, int MF, int j, float sample_dist) {
  BgGeom gq;
  gq.dist = j + 1 < MF ? zf[j + 1] - zf[j] : sample_dist;
  gq.mid = zf[j] + gq.dist * 0.5f;
  for (int c = 0; c < 3; ++c) gq.pts[c] = o[c] + d[c] * gq.mid;
  gq.rn = sqrtf(gq.pts[0] * gq.pts[0] + gq.pts[1] * gq.pts[1] + gq.pts[2] * gq.pts[2]);
  gq.clipped = !(gq.rn >= 1.0f && gq.rn <= 1e10f);
  gq.r = fminf(fmaxf(gq.rn, 1.0f), 1e10f);
  for (int c = 0; c < 3; ++c) gq.q[c] = gq.pts[c] / gq.r;
  gq.q[3] = 1.0f / gq.r;
  return gq;
}
CNR_HD void body_bg_embed(const BgEmbed& p, long idx) {
  const long ray = idx / p.MF;
  const int j = (int)(idx - ray * p.MF);
  const float* d = p.d + ray * 3;
  const BgGeom gq = bg_geometry(p.o + ray * 3, d, p.z_feed + ray * p.MF, p.MF, j, p.sample_dist);
  p.dist[idx] = gq.dist;
  const int ne = 4 + 8 * p.multires;
  float* e = p.E + idx * p.lde;
  for (int c = 0; c < 4; ++c) e[c] = gq.q[c];
  float f = 1.0f;
  for (int k = 0; k < p.multires; ++k) {
    for (int c = 0; c < 4; ++c) {
      float sn, cs;
      sincosf(gq.q[c] * f, &sn, &cs);
      e[4 + 8 * k + c] = sn;
      e[8 + 8 * k + c] = cs;
    }
    f *= 2.0f;
  }
  for (int c = ne; c < p.lde; ++c) e[c] = 0.0f;
  if (p.XH) {
    float* x = p.XH + idx * p.ldxh;
    for (int c = 0; c < ne; ++c) x[p.xh_off + c] = e[c];
    for (int c = p.xh_off + ne; c < p.ldxh; ++c) x[c] = 0.0f;
  }
  float* v = p.FV + idx * p.ldfv + p.fv_off;
  const int nv = 3 + 6 * p.multires_view;
  pe_row(d, p.multires_view, v);
  for (int c = p.fv_off + nv; c < p.ldfv; ++c) p.FV[idx * p.ldfv + c] = 0.0f;
}
CNR_HD float softplus1(float x) { return x > 20.0f ? x : log1pf(expf(x)); }    // F.softplus (beta 1, threshold 20)
CNR_HD void body_bg_alpha(const BgAlpha& p, long i) { p.alpha[i] = 1.0f - expf(-softplus1(p.density[i]) * p.dist[i]); }
CNR_HD void body_bg_heads_bwd(const BgHeadsBwd& p, long i) {
  const float x = p.density[i], sp = softplus1(x), e = expf(-sp * p.dist[i]), da = p.d_alpha ? p.d_alpha[i] : 0.0f;
  const float dsp = da * e * p.dist[i];
  float* dd = p.d_density + i * p.ldd;
  dd[0] = x > 20.0f ? dsp : dsp * sigmoidf_(x);
  for (int c = 1; c < p.ldd; ++c) dd[c] = 0.0f;
  p.d_dist[i] = da * e * sp;
  float* dr = p.d_rgb_pre + i * p.ldr;
  for (int c = 0; c < 3; ++c) { const float y = p.rgb[i * 3 + c]; dr[c] = (p.d_rgb ? p.d_rgb[i * 3 + c] : 0.0f) * y * (1.0f - y); }
  for (int c = 3; c < p.ldr; ++c) dr[c] = 0.0f;
}
CNR_HD void body_bg_join(const BgJoin& p, long e) {
  const long pt = e / p.W;
  const int c = (int)(e - pt * p.W);
  p.dZ[e] = p.H[e] > 0.0f ? p.T[e] + p.d_density[pt * p.ldd] * p.w_alpha[c] : 0.0f;
}
CNR_HD void body_bg_embed_bwd(const BgEmbedBwd& p, long idx) {
  const long ray = idx / p.MF;
  const int j = (int)(idx - ray * p.MF);
  const float* d = p.d + ray * 3;
  const BgGeom gq = bg_geometry(p.o + ray * 3, d, p.z_feed + ray * p.MF, p.MF, j, p.sample_dist);
  const float* e0 = p.dE0 + idx * p.lde0;
  const float* e1 = p.dE1 ? p.dE1 + idx * p.lde1 : nullptr;
  float dq[4];
  for (int c = 0; c < 4; ++c) dq[c] = e0[c] + (e1 ? e1[c] : 0.0f);
  float f = 1.0f;
  for (int k = 0; k < p.multires; ++k) {
    for (int c = 0; c < 4; ++c) {
      float sn, cs;
      sincosf(gq.q[c] * f, &sn, &cs);
      const float ds = e0[4 + 8 * k + c] + (e1 ? e1[4 + 8 * k + c] : 0.0f), dc = e0[8 + 8 * k + c] + (e1 ? e1[8 + 8 * k + c] : 0.0f);
      dq[c] += f * (cs * ds - sn * dc);
    }
    f *= 2.0f;
  }
  float* out = p.dp + idx * 8;
  if (gq.clipped) {            // r is the clip bound: a constant
    for (int c = 0; c < 3; ++c) out[c] = dq[c] / gq.r;
  } else {                     // q = p / |p|, s = 1 / |p|
    const float qd = gq.q[0] * dq[0] + gq.q[1] * dq[1] + gq.q[2] * dq[2];
    const float r3 = gq.r * gq.r * gq.r;
    for (int c = 0; c < 3; ++c) out[c] = (dq[c] - gq.q[c] * qd) / gq.r - dq[3] * gq.pts[c] / r3;
  }
  const float* ve = p.dVE + idx * p.ldve;
  float dv[3] = {ve[0], ve[1], ve[2]};
  f = 1.0f;
  for (int k = 0; k < p.multires_view; ++k) {
    for (int c = 0; c < 3; ++c) {
      float sn, cs;
      sincosf(d[c] * f, &sn, &cs);
      dv[c] += f * (cs * ve[3 + 6 * k + c] - sn * ve[6 + 6 * k + c]);
    }
    f *= 2.0f;
  }
  out[3] = dv[0]; out[4] = dv[1]; out[5] = dv[2]; out[6] = 0.0f; out[7] = 0.0f;
}
CNR_HD void body_bg_rays_bwd(const BgRaysBwd& p, long ray) {
  const int MF = p.MF;
  const float* d = p.d + ray * 3;
  const float* zf = p.z_feed + ray * MF;
  float dO[3] = {0.f, 0.f, 0.f}, dD[3] = {0.f, 0.f, 0.f};
  float carry = 0.0f;                                          // d dist_{j-1}: enters d z_j with a plus sign
  for (int j = 0; j < MF; ++j) {
    const float* q = p.dp + (ray * MF + j) * 8;
    const float dist = j + 1 < MF ? zf[j + 1] - zf[j] : p.sample_dist;
    const float mid = zf[j] + dist * 0.5f;
    const float dmid = q[0] * d[0] + q[1] * d[1] + q[2] * d[2];
    for (int c = 0; c < 3; ++c) { dO[c] += q[c]; dD[c] += q[c] * mid + q[3 + c]; }
    const float ddist = j + 1 < MF ? p.d_dist[ray * MF + j] + 0.5f * dmid : 0.0f;   // (the last section length is a constant)
    p.d_z_feed[ray * MF + j] += dmid - ddist + carry;
    carry = ddist;
  }
  for (int c = 0; c < 3; ++c) { p.d_o[ray * 3 + c] = dO[c]; p.d_d[ray * 3 + c] = dD[c]; }
}

// one foreground section of render_core (Color_NeuS.py:41-90 / NeuS.py:209-256)
struct FgSample { float dist, inside, relax, gn, g[3]; AlphaOut a; };
CNR_HD FgSample fg_sample(const CompositeBg& p, long ray, int j, float inv_s) {
  FgSample q;
  const float* z = p.z + ray * p.M;
  const float* o = p.o + ray * 3;
  const float* d = p.d + ray * 3;
  q.dist = j + 1 < p.M ? z[j + 1] - z[j] : p.sample_dist;
  const float mid = z[j] + q.dist * 0.5f;
  const float x = o[0] + d[0] * mid, y = o[1] + d[1] * mid, w = o[2] + d[2] * mid;
  const float pn = sqrtf(x * x + y * y + w * w);
  q.inside = pn < 1.0f ? 1.0f : 0.0f;
  q.relax = pn < 1.2f ? 1.0f : 0.0f;
  const long pt = ray * p.M + j;
  for (int c = 0; c < 3; ++c) q.g[c] = p.g[pt * 3 + c];
  q.gn = sqrtf(q.g[0] * q.g[0] + q.g[1] * q.g[1] + q.g[2] * q.g[2]);
  q.a = alpha_forward(p.sdf[pt], q.g, d, q.dist, inv_s, p.cos_anneal);
  return q;
}
CNR_HD void body_composite_bg(const CompositeBg& p, long ray) {
  const int M = p.M, MF = p.MF;
  const float inv_s = fminf(fmaxf(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);
  float T = 1.0f, Tin = 1.0f, wsum = 0.f, wmax = -1.f, dep = 0.f, col[3] = {0.f, 0.f, 0.f}, gcl[3] = {0.f, 0.f, 0.f}, e0 = 0.f, e1 = 0.f;
  for (int j = 0; j < MF; ++j) {
    float alpha = p.bg_alpha[ray * MF + j];
    float c3[3] = {p.bg_color[(ray * MF + j) * 3], p.bg_color[(ray * MF + j) * 3 + 1], p.bg_color[(ray * MF + j) * 3 + 2]};
    if (j < M) {
      const FgSample q = fg_sample(p, ray, j, inv_s);
      const long pt = ray * M + j;
      alpha = q.a.alpha * q.inside + alpha * (1.0f - q.inside);
      for (int k = 0; k < 3; ++k) c3[k] = p.color[pt * 3 + k] * q.inside + c3[k] * (1.0f - q.inside);
      const float w_in = q.a.alpha * Tin;
      Tin = Tin * (1.0f - q.a.alpha + 1e-7f);
      if (p.gcolor) for (int k = 0; k < 3; ++k) gcl[k] += w_in * p.gcolor[pt * 3 + k];
      e0 += q.relax * (q.gn - 1.0f) * (q.gn - 1.0f);
      e1 += q.relax;
      p.cdf_fine[pt] = q.a.pc;
      p.inside_sphere[pt] = q.inside;
    }
    const float w = alpha * T;
    T = T * (1.0f - alpha + 1e-7f);
    p.weights[ray * MF + j] = w;
    wsum += w; wmax = fmaxf(wmax, w); dep += w * p.z_feed[ray * MF + j];
    for (int k = 0; k < 3; ++k) col[k] += w * c3[k];
  }
  for (int k = 0; k < 3; ++k) {
    p.color_fine[ray * 3 + k] = p.background_rgb ? col[k] + p.background_rgb[k] * (1.0f - wsum) : col[k];
    if (p.global_color) p.global_color[ray * 3 + k] = gcl[k];
  }
  p.weight_sum[ray] = wsum; p.weight_max[ray] = wmax; p.depth[ray] = dep; p.s_val[ray] = 1.0f / inv_s;
  p.eik_partial[ray * 2] = e0; p.eik_partial[ray * 2 + 1] = e1;
}
CNR_HD void body_composite_bg_bwd(const CompositeBgBwd& b, long ray) {
  const CompositeBg& p = b.f;
  const int M = p.M, MF = p.MF;
  const float inv_s = fminf(fmaxf(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);
  const float* d = p.d + ray * 3;
  const float* w = p.weights + ray * MF;
  // per-ray cotangents
  float dcol[3] = {0.f, 0.f, 0.f}, dgc[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < 3; ++k) { if (b.d_color_fine) dcol[k] = b.d_color_fine[ray * 3 + k]; if (b.d_global_color && p.gcolor) dgc[k] = b.d_global_color[ray * 3 + k]; }
  float dwsum = b.d_weight_sum ? b.d_weight_sum[ray] : 0.0f;
  if (p.background_rgb) for (int k = 0; k < 3; ++k) dwsum -= dcol[k] * p.background_rgb[k];
  const float ddep = b.d_depth ? b.d_depth[ray] : 0.0f;
  const float dwmax = b.d_weight_max ? b.d_weight_max[ray] : 0.0f;
  int jmax = 0;
  { float best = w[0]; for (int j = 1; j < MF; ++j) if (w[j] > best) { best = w[j]; jmax = j; } }
  // eikonal: gradient_error = E0 / (E1 + 1e-5)
  const float dge = b.d_gradient_error ? b.d_gradient_error[0] : 0.0f;
  const float dE0 = dge / (b.eik_sums[1] + 1e-5f);
  // forward quantities again, in double (the transmittances in scan order; the weights are re-formed from them rather than read back in float32)
  double T[kMaxRaySamples], Tin[kMaxRaySamples], alpha_m[kMaxRaySamples];
  {
    double t = 1.0, ti = 1.0;
    for (int j = 0; j < MF; ++j) {
      double alpha = p.bg_alpha[ray * MF + j];
      if (j < M) {
        const FgSample q = fg_sample(p, ray, j, inv_s);
        const AlphaOutD ad = alpha_forward_d(p.sdf[ray * M + j], q.g, d, q.dist, inv_s, p.cos_anneal);
        const double a_in = fmin(fmax(ad.a_raw, 0.0), 1.0);
        alpha = a_in * q.inside + alpha * (1.0 - q.inside);
        Tin[j] = ti; ti = ti * (1.0 - a_in + 1e-7);
      }
      alpha_m[j] = alpha; T[j] = t; t = t * (1.0 - alpha + 1e-7);
    }
  }
  double S = 0.0, Sin = 0.0;                                   // suffix sums of d w_k * w_k (double: one thread per ray, and the terms cancel)
  float drd[3] = {0.f, 0.f, 0.f};
  double dinvs = 0.0;                                          // (terms of both signs over the whole ray: summed in double, one thread per ray)
  for (int j = MF - 1; j >= 0; --j) {
    const long fj = ray * MF + j;
    float bc[3] = {p.bg_color[fj * 3], p.bg_color[fj * 3 + 1], p.bg_color[fj * 3 + 2]};
    float c3[3] = {bc[0], bc[1], bc[2]};
    FgSample q;
    q.inside = 0.0f;
    if (j < M) {
      q = fg_sample(p, ray, j, inv_s);
      for (int k = 0; k < 3; ++k) c3[k] = p.color[(ray * M + j) * 3 + k] * q.inside + bc[k] * (1.0f - q.inside);
    }
    // total cotangent of w_j
    const double wj = alpha_m[j] * T[j];
    double dw = (double)(b.d_weights ? b.d_weights[fj] : 0.0f) + dwsum + (double)ddep * p.z_feed[fj] + (j == jmax ? dwmax : 0.0f);
    for (int k = 0; k < 3; ++k) dw += (double)dcol[k] * c3[k];
    const double dalpha = dw * T[j] - S / (1.0 - alpha_m[j] + 1e-7);
    S += dw * wj;
    b.d_z_feed[fj] = ddep * w[j];
    // colours
    for (int k = 0; k < 3; ++k) b.d_bg_color[fj * 3 + k] = dcol[k] * w[j] * (j < M ? 1.0f - q.inside : 1.0f);
    if (j >= M) { b.d_bg_alpha[fj] = (float)dalpha; continue; }
    const long pt = ray * M + j;
    b.d_bg_alpha[fj] = (float)(dalpha * (1.0 - q.inside));
    for (int k = 0; k < 3; ++k) b.d_color[pt * 3 + k] = dcol[k] * w[j] * q.inside;
    // foreground-only weights of the global colour
    const AlphaOutD ad = alpha_forward_d(p.sdf[pt], q.g, d, q.dist, inv_s, p.cos_anneal);
    const double a_in = fmin(fmax(ad.a_raw, 0.0), 1.0);
    double da_in = 0.0;
    if (p.gcolor) {
      const double w_in = a_in * Tin[j];
      double dwi = 0.0;
      for (int k = 0; k < 3; ++k) { dwi += (double)dgc[k] * p.gcolor[pt * 3 + k]; b.d_gcolor[pt * 3 + k] = (float)(dgc[k] * w_in); }
      da_in = dwi * Tin[j] - Sin / (1.0 - a_in + 1e-7);
      Sin += dwi * w_in;
    }
    double dinv1 = 0.0;
    const AlphaGrad ag = alpha_backward_d(ad, q.dist, inv_s, p.cos_anneal, dalpha * q.inside + da_in, b.d_cdf ? b.d_cdf[pt] : 0.0f, &dinv1);
    b.d_sdf[pt] = ag.d_sdf;
    dinvs += dinv1;
    const float dgn = q.relax * 2.0f * (q.gn - 1.0f) * dE0;
    for (int c = 0; c < 3; ++c) {
      float dg = ag.d_tc * d[c] + (q.gn > 0.0f ? dgn * q.g[c] / q.gn : 0.0f);
      if (b.d_gradients) dg += b.d_gradients[pt * 3 + c];
      b.d_g[pt * 3 + c] = dg;
      drd[c] += ag.d_tc * q.g[c];
    }
    // dist_j = z_{j+1} - z_j (j < M - 1): d z_j -= d dist_j, d z_{j+1} += d dist_j
    const float ddist = j + 1 < M ? ag.d_dist : 0.0f;
    if (b.d_z) {
      if (j + 1 < M) b.d_z[pt + 1] += ddist;                   // (row j + 1 was written in the previous iteration)
      b.d_z[pt] = -ddist;
    }
  }
  // s_val = 1 / inv_s for every sample of the ray
  if (b.d_s_val) dinvs += (double)(-b.d_s_val[ray] / (inv_s * inv_s));
  b.d_inv_s_partial[ray] = (float)dinvs;                              // (the clip of inv_s is applied by the variance reduction, be_variance_finish)
  for (int c = 0; c < 3; ++c) b.d_rays_d[ray * 3 + c] = drd[c];
}

}  // namespace cnr
