// CPU emulation of the kernel layer -- TEST INFRASTRUCTURE ONLY (built with -DCNR_CPU_EMU into
// libcolorneus_emu.so, loaded only by tests/ to exercise the host orchestration, operand views, epilogues and
// per-point bodies without a GPU).  The product path never loads this library.
//
// GEMMs are plain loops over the same views / epilogues; per-ray kernels are straightforward serial code.
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "cnr_backend.h"
#include "cnr_mc_table.h"
#include "cnr_loss.h"
#include "cnr_bodies.h"

namespace cnr {

const char* be_name() { return "cpu-emu"; }
void be_split_planes(const float*, int, int, unsigned short*, float*, cnr_stream) {}
void be_split_planes_many(const SplitJob*, int, cnr_stream) {}
void be_prep_weights(const PrepWeight* p, int count, cnr_stream s) { for (int i = 0; i < count; ++i) be_prep_weight(p[i], s); }
void be_finish_weights(const FinishWeight* f, int count, cnr_stream s) { for (int i = 0; i < count; ++i) be_finish_weight(f[i], s); }
void be_timing_enable(int) {}
int be_timing_collect(KernelTiming*, int) { return 0; }
int be_check_last_error(char*, size_t) { return 0; }
void be_range_push(const char*) {}
void be_range_pop() {}
void be_memset_zero(void* p, size_t bytes, cnr_stream) { memset(p, 0, bytes); }
void be_loss_sums(const LossArgs& a, float* partial, float* sums, cnr_stream) {
  (void)partial;
  double s_rgb = 0, s_bce = 0, s_rel = 0;   // (the emulation only has to agree with the oracle to test tolerance, not bitwise with the GPU)
  for (long i = 0; i < a.R * 3; ++i) s_rgb += loss_rgb_term(a.color[i], a.gt[i], a.rgb_l1);
  if (a.mask) for (long r = 0; r < a.R; ++r) s_bce += loss_bce_term(a.wsum[r], a.mask[r]);
  if (a.drel) {
    const long per_ray = a.drel_per_ray ? 1 : (long)a.M * 3;
    for (long i = 0; i < a.R * per_ray; ++i) s_rel += a.drel[i] * ((a.include_mask && a.mask) ? a.mask[i / per_ray] : 1.0f);
  }
  sums[0] = (float)s_rgb; sums[1] = (float)s_bce; sums[2] = (float)s_rel; sums[3] = 0.f;
}
void be_loss_combine(const LossScalars& c, const float* sums, const float* gerr, float* out6, cnr_stream) { loss_combine(c, sums, gerr, out6); }
void be_loss_forward(const LossArgs& a, float* partial, unsigned* ticket, const LossScalars& c, const float* gerr, float* sums, float* out6, cnr_stream s) {
  (void)ticket;
  be_loss_sums(a, partial, sums, s);
  loss_combine(c, sums, gerr, out6);
}
void be_loss_shard_stats(const LossArgs& a, float* partial, unsigned* ticket, const float* eik_sums, float* stats8, cnr_stream s) {
  (void)ticket;
  float sums[4];
  be_loss_sums(a, partial, sums, s);
  stats8[0] = sums[0]; stats8[1] = sums[1]; stats8[2] = sums[2]; stats8[3] = eik_sums[0]; stats8[4] = eik_sums[1]; stats8[5] = eik_sums[1]; stats8[6] = stats8[7] = 0.0f;
}
void be_loss_shard_combine(const LossScalars& c, const float* stats8, float* out8, cnr_stream) { loss_shard_combine(c, stats8, out8); }
void be_loss_backward(const LossArgs& a, const LossScalars& c, const float* g_loss, const float* mean_rel, const float* eik_factor, float* coef4, float* d_color,
                      float* d_wsum, float* d_drel_ray, cnr_stream s) {
  loss_coef(c, g_loss, c.use_relight ? mean_rel : g_loss, eik_factor, coef4);
  be_loss_grads(a, coef4, d_color, d_wsum, nullptr, s);
  if (d_drel_ray) for (long r = 0; r < a.R; ++r) d_drel_ray[r] = (a.include_mask && a.mask) ? coef4[2] * a.mask[r] : coef4[2];
}
void be_loss_coef(const LossScalars& c, const float* g_loss, const float* mean_rel, float* coef4, cnr_stream) { loss_coef(c, g_loss, mean_rel, nullptr, coef4); }
void be_loss_grads(const LossArgs& a, const float* coef, float* d_color, float* d_wsum, float* d_drel, cnr_stream) {
  for (long i = 0; i < a.R * 3; ++i) d_color[i] = coef[0] * loss_rgb_grad(a.color[i], a.gt[i], a.rgb_l1);
  if (d_wsum) for (long r = 0; r < a.R; ++r) d_wsum[r] = a.mask ? coef[1] * loss_bce_grad(a.wsum[r], a.mask[r]) : 0.0f;
  if (d_drel) {
    const long per_ray = (long)a.M * 3;
    for (long i = 0; i < a.R * per_ray; ++i) d_drel[i] = coef[2] * ((a.include_mask && a.mask) ? a.mask[i / per_ray] : 1.0f);
  }
}
void be_zero_cols(float* p, int ld, int c0, int c1, long rows, cnr_stream) {
  for (long r = 0; r < rows; ++r) for (int c = c0; c < c1; ++c) p[r * ld + c] = 0.0f;
}
void be_grid_points(float*, cnr_stream) {}
void be_copy_cols(float* dst, int ld_dst, const float* src, int ld_src, int ncols, long rows, cnr_stream) {
  for (long r = 0; r < rows; ++r) memcpy(dst + r * ld_dst, src + r * ld_src, (size_t)ncols * sizeof(float));
}

void be_layer_gemm(const LayerGemm& g0, cnr_stream) {
  LayerGemm g = g0;
  if (g.P_dev) g.P = *g.P_dev;
  g.K += g.k_extra;   // (the rank-one form is a speed matter of the fused HIP launch)
  const int kp = round_up(g.K, 4);
  std::vector<float> arow(kp + 4);
#pragma omp parallel for firstprivate(arow)
  for (long row = 0; row < g.P; ++row) {
    for (int k = 0; k < kp; k += 4) {
      f4 v = view_eval4(g.A, row, k);
      arow[k] = v.x; arow[k + 1] = v.y; arow[k + 2] = v.z; arow[k + 3] = v.w;
    }
    if (g.dot_w && g.first_col == 0) {   // row dot (LayerGemm::dot_*): the sdf row of the top SDF layer
      float acc = 0.0f;
      for (int k = 0; k < g.K; ++k) acc = fmaf(arow[k], g.dot_w[k], acc);
      g.dot_out[row] = (acc + (g.dot_bias ? g.dot_bias[0] : 0.0f)) * g.dot_scale;
    }
    int ncols = g.N;
    if (g.E.tail_src && g.E.n_out + g.E.tail_n > ncols) ncols = g.E.n_out + g.E.tail_n;
    for (int n = g.first_col; n < round_up(ncols, 32); ++n) {
      const float* w = g.W + (long)(n < round_up(g.N, 32) ? n : 0) * g.ldw;
#ifdef CNR_EMU_DOUBLE_ACC
      double accd = 0.0;
      if (n < round_up(g.N, 32)) for (int k = 0; k < g.K; ++k) accd += (double)arow[k] * (double)w[k];
      float acc = (float)accd;
#else
      float acc = 0.0f;
      if (n < round_up(g.N, 32)) for (int k = 0; k < g.K; ++k) acc = fmaf(arow[k], w[k], acc);
#endif
      epi_apply(g.E, row, n, acc);
    }
  }
}

void be_dw_gemm(const DwGemm& g, cnr_stream) {
#pragma omp parallel for
  for (int chunk = 0; chunk < g.nchunk; ++chunk) {
    float* out = g.partial + (long)chunk * g.Npad * g.ldk;
    std::fill(out, out + (long)g.Npad * g.ldk, 0.0f);
    if (g.colsum) std::fill(g.colsum + (long)chunk * g.Npad, g.colsum + (long)(chunk + 1) * g.Npad, 0.0f);
    long p0 = chunk * g.chunk_pts, p1 = std::min(g.P, p0 + g.chunk_pts);
    std::vector<float> x(round_up(g.N, 4) + 4), y(round_up(g.K, 4) + 4);
    for (int pair = 0; pair < g.npairs; ++pair)
      for (long pt = p0; pt < p1; ++pt) {
        for (int n = 0; n < g.N; n += 4) { f4 v = view_eval4(g.X[pair], pt, n); x[n] = v.x; x[n + 1] = v.y; x[n + 2] = v.z; x[n + 3] = v.w; }
        for (int k = 0; k < g.K; k += 4) { f4 v = view_eval4(g.Y[pair], pt, k); y[k] = v.x; y[k + 1] = v.y; y[k + 2] = v.z; y[k + 3] = v.w; }
        for (int n = 0; n < g.N; ++n) {   // (skip_main is a speed matter of the HIP backend: the emulation forms every tile here)
          float xv = x[n];
          if (pair == 0 && g.colsum) g.colsum[(long)chunk * g.Npad + n] += xv;
          if (xv == 0.0f) continue;
          float* o = out + (long)n * g.ldk;
          for (int k = 0; k < g.K; ++k) o[k] += xv * y[k];
        }
      }
  }
}

#define CNR_PW(NAME, PARAM, BODY, COUNT)                \
  void NAME(const PARAM& p, cnr_stream) {               \
    const long n_ = (COUNT);                            \
    _Pragma("omp parallel for") for (long i = 0; i < n_; ++i) BODY(p, i); \
  }
CNR_PW(be_embed_z, EmbedZ, body_embed_z, p.R* p.m)
CNR_PW(be_embed_pts, EmbedPts, body_embed_pts, p.n)
CNR_PW(be_fine_setup, FineSetup, body_fine_setup, p.R* p.M)
CNR_PW(be_grad_finish, GradFinish, body_grad_finish, p.P)
CNR_PW(be_coltop_bwd, ColTopBwd, body_coltop_bwd, p.P)
CNR_PW(be_gbar_finish, GbarFinish, body_gbar_finish, p.P)
CNR_PW(be_pbar_finish, PbarFinish, body_pbar_finish, p.P)
CNR_PW(be_outside_z, OutsideZ, body_outside_z, p.R)
CNR_PW(be_outside_z_bwd, OutsideZBwd, body_outside_z_bwd, p.R)
CNR_PW(be_bg_embed, BgEmbed, body_bg_embed, p.R* p.MF)
CNR_PW(be_bg_alpha, BgAlpha, body_bg_alpha, p.n)
CNR_PW(be_bg_heads_bwd, BgHeadsBwd, body_bg_heads_bwd, p.n)
CNR_PW(be_bg_join, BgJoin, body_bg_join, p.n* p.W)
CNR_PW(be_bg_embed_bwd, BgEmbedBwd, body_bg_embed_bwd, p.R* p.MF)
CNR_PW(be_bg_rays_bwd, BgRaysBwd, body_bg_rays_bwd, p.R)
CNR_PW(be_composite_bg, CompositeBg, body_composite_bg, p.R)
CNR_PW(be_composite_bg_bwd, CompositeBgBwd, body_composite_bg_bwd, p.f.R)

void be_head_bwd(const HeadBwd& p, cnr_stream) {
  const long per = round_up((int)((p.P + p.nslots - 1) / p.nslots), 64);
#pragma omp parallel for
  for (int slot = 0; slot < p.nslots; ++slot) {
    float* out = p.partial + (long)slot * p.npad * p.ldk;
    for (int j = 0; j < p.n; ++j) {
      for (int k = 0; k < p.ldk; ++k) out[(long)j * p.ldk + k] = 0.0f;
      if (p.colsum) p.colsum[(long)slot * p.npad + j] = 0.0f;
    }
    const long p0 = slot * per, p1 = std::min(p.P, p0 + per);
    for (long pt = p0; pt < p1; ++pt)
      for (int k = 0; k < p.K; ++k) {
        const float a = p.aux[pt * p.ldaux + k];
        float v = 0.0f;
        for (int j = 0; j < p.n; ++j) {
          const float d = p.dtop[pt * p.ldt + j];
          v = fmaf(p.W[(long)j * p.ldw + k], d, v);
          out[(long)j * p.ldk + k] = fmaf(d, a, out[(long)j * p.ldk + k]);
          if (k == 0 && p.colsum) p.colsum[(long)slot * p.npad + j] += d;
        }
        p.dout[pt * p.ldo + k] = a > 0.0f ? v : 0.0f;
      }
  }
}
void be_head_fwd(const HeadFwd& p, cnr_stream) {
  const long P = p.P_dev ? (long)*p.P_dev : p.P;
#pragma omp parallel for
  for (long row = 0; row < P; ++row)
    for (int j = 0; j < 16; ++j) {
      float acc = 0.0f;
      if (j < p.n) for (int c = 0; c < 256; ++c) acc = fmaf(p.h[row * p.ldh + c], p.W[(long)j * p.ldw + c], acc);
      epi_apply(p.E, row, j, acc);
    }
}
void be_sampler_step(const SamplerStep& p, cnr_stream s) {   // the three steps one after the other (the fused launch is a speed matter)
  if (p.do_merge) be_merge(p.g, s);
  be_upsample(p.u, s);
  if (p.do_embed) {
    EmbedZ e;
    e.o = p.u.o; e.d = p.u.d; e.R = p.u.R; e.m = p.u.m; e.z = p.u.new_z; e.ldz = p.u.m; e.make_z = 0;
    e.near_ = nullptr; e.far_ = nullptr; e.t_rand = nullptr; e.scale = p.scale; e.multires = p.multires; e.E = p.E;
    be_embed_z(e, s);
  }
}
void be_strip_bwd(const StripBwd& p, cnr_stream) {
  const long per = round_up((int)((p.P + p.nslots - 1) / p.nslots), 64);
#pragma omp parallel for
  for (int slot = 0; slot < p.nslots; ++slot) {
    float* out = p.partial + (long)slot * p.npad * p.ldk;
    for (int n = 0; n < 256 && n < p.npad; ++n)
      for (int k = 256; k < p.ldk; ++k) out[(long)n * p.ldk + k] = 0.0f;
    const long p0 = slot * per, p1 = std::min(p.P, p0 + per);
    for (long pt = p0; pt < p1; ++pt)
      for (int j = 0; j < p.nt; ++j) {
        const float y = p.y[pt * p.ldy + j];
        float dot = 0.0f;
        for (int n = 0; n < 256; ++n) {
          const float d = p.dout[pt * p.ldo + n];
          dot = fmaf(d, p.Wt[(long)(256 + j) * p.ldwt + n], dot);
          if (n < p.npad) out[(long)n * p.ldk + 256 + j] = fmaf(d, y, out[(long)n * p.ldk + 256 + j]);
        }
        if (p.tail) p.tail[pt * p.ldt + j] = dot * p.tail_scale;
      }
  }
}
bool be_fdw_enabled() { return !debug_flags().no_fdw; }
bool be_fdw_xrow() { return be_fdw_enabled(); }
void be_layer_dw_gemm(const LayerGemm& g, const DwGemm& d, const DwFuse& f, cnr_stream s) {
  // xrow_mode 2: the plain weight-gradient GEMM below forms the extra row itself (d.N covers it) but zero-fills the slots first: keep what
  // the mode-1 launch left there and add it back
  std::vector<float> keep;
  if (f.xrow_mode == 2) {
    keep.resize((size_t)f.nslots * 256);
    for (int c = 0; c < f.nslots; ++c) memcpy(&keep[(size_t)c * 256], f.xrow + (long)c * f.xrow_stride, 256 * sizeof(float));
  }
  be_dw_gemm(d, s);      // (first: EK_VBACK updates o1 in place, but neither dW operand is an output of this launch, so the order is free)
  be_layer_gemm(g, s);
  if (f.xrow_mode == 2)
    for (int c = 0; c < f.nslots; ++c)
      for (int k = 0; k < 256; ++k) f.xrow[(long)c * f.xrow_stride + k] += keep[(size_t)c * 256 + k];
  if (f.xrow_mode == 1)   // column sums of the o2 output over the point slices of d
    for (int c = 0; c < f.nslots; ++c) {
      const long p0 = c * d.chunk_pts, p1 = std::min(d.P, p0 + d.chunk_pts);
      for (int k = 0; k < 256; ++k) {
        float sum = 0.0f;
        for (long pt = p0; pt < p1; ++pt) sum += g.E.o2[pt * g.E.ld2 + k];
        f.xrow[(long)c * f.xrow_stride + k] = sum * f.xrow_scale;
      }
    }
}

static int seg_src_of(const Segment* seg, int nseg, int j) {
  for (int q = 0; q < nseg; ++q)
    if (j >= seg[q].dst && j < seg[q].dst + seg[q].len) return seg[q].src + (j - seg[q].dst);
  return -1;
}

void be_prep_weight(const PrepWeight& p, cnr_stream) {
  for (int n = 0; n < p.npad; ++n) {
    const bool real = n < p.n;
    const int nr = real ? (n + p.row_rot) % p.n : 0;
    float scale = 1.0f;
    if (p.g && real) {
      double ss = 0.0;   // row norm accumulated in double (weight_norm is the most rounding-sensitive step: x inv_s downstream)
      for (int c = 0; c < p.k_ref; ++c) { double x = p.v[(long)nr * p.k_ref + c]; ss += x * x; }
      scale = p.g[nr] / (float)sqrt(ss);
    }
    for (int j = 0; j < p.kpad; ++j) {
      float val = 0.0f;
      if (real && j < p.ldw) {
        int src = seg_src_of(p.seg, p.nseg, j);
        if (src >= 0) val = p.v[(long)nr * p.k_ref + src] * scale;
      }
      if (j < p.ldw) p.W[(long)n * p.ldw + j] = val;
      if (n < p.ldwt) p.Wt[(long)j * p.ldwt + n] = val;
    }
    p.bias[n] = real && p.b ? p.b[nr] : 0.0f;
  }
}

void be_finish_weight(const FinishWeight& p, cnr_stream) {
  std::vector<float> dwi(p.ldk), dref(p.k_ref);
  for (int n = 0; n < p.n; ++n) {
    const int nr = (n + p.row_rot) % p.n;
    for (int j = 0; j < p.ldk; ++j) {
      float s = 0.0f;
      const int nch = j >= p.col_hi ? p.nchunk_hi : p.nchunk;
      for (int c = 0; c < nch; ++c) s += p.partial[((long)c * p.npad + n) * p.ldk + j];
      dwi[j] = s;
    }
    for (int c = 0; c < p.k_ref; ++c) {
      int dst = -1;
      for (int q = 0; q < p.nseg; ++q)
        if (c >= p.seg[q].src && c < p.seg[q].src + p.seg[q].len) dst = p.seg[q].dst + (c - p.seg[q].src);
      dref[c] = dst >= 0 ? dwi[dst] : 0.0f;
    }
    if (p.g) {
      double dotd = 0.0, ssd = 0.0;
      for (int c = 0; c < p.k_ref; ++c) { double vv = p.v[(long)nr * p.k_ref + c]; dotd += (double)dref[c] * vv; ssd += vv * vv; }
      float dot = (float)dotd, nrm = (float)sqrt(ssd), gg = p.g[nr];
      p.dg[nr] = dot / nrm;
      for (int c = 0; c < p.k_ref; ++c) {
        float vv = p.v[(long)nr * p.k_ref + c];
        p.dv[(long)nr * p.k_ref + c] = (gg / nrm) * (dref[c] - dot / (nrm * nrm) * vv);
      }
    } else {
      for (int c = 0; c < p.k_ref; ++c) p.dv[(long)nr * p.k_ref + c] = dref[c];
    }
    if (p.db && p.colsum) {
      float s = 0.0f;
      for (int c = 0; c < (p.ncolsum > 0 ? p.ncolsum : p.nchunk); ++c) s += p.colsum[(long)c * p.npad + n];
      p.db[nr] = s;
    }
  }
}

void be_reduce_eik(const ReduceEik& p, cnr_stream) {
  float a = 0.0f, b = 0.0f;
  for (long r = 0; r < p.R; ++r) { a += p.partial[r * 2]; b += p.partial[r * 2 + 1]; }
  p.sums[0] = a; p.sums[1] = b;
  if (p.sums_out) { p.sums_out[0] = a; p.sums_out[1] = b; }
  *p.gradient_error = a / (b + 1e-5f);
}

void be_variance_finish(const VarianceFinish& p, cnr_stream) {
  float a = 0.0f;
  for (long r = 0; r < p.R; ++r) a += p.partial[r];
  float raw = expf(p.variance[0] * 10.0f);
  *p.d_variance = (raw >= 1e-6f && raw <= 1e6f) ? a * 10.0f * raw : 0.0f;
}

void be_mc_count(const McVolume& m, cnr_stream) {
  const int res = m.res;
  const long n = (long)res * res * res;
  int vo = 0, to = 0;
  for (long v = 0; v < n; ++v) {
    const int z = (int)(v % res), y = (int)((v / res) % res), x = (int)(v / ((long)res * res));
    unsigned char fl;
    const int idx = mc_cell(m.u, res, m.thr, x, y, z, &fl);
    m.flags[v] = fl;
    m.counts[v * 2] = vo; m.counts[v * 2 + 1] = to;
    vo += mc_popcount3(fl);
    to += idx >= 0 ? kMcNumTris[idx] : 0;
  }
  m.totals[0] = vo; m.totals[1] = to;
}
void be_mc_emit(const McVolume& m, const float* bmin, const float* bmax, float* verts, int* tris, cnr_stream) {
  const int res = m.res;
  const long r2 = (long)res * res, n = r2 * res;
  const long stride[3] = {r2, (long)res, 1};
  for (long v = 0; v < n; ++v) {
    const unsigned fl = m.flags[v];
    const int xyz[3] = {(int)(v / r2), (int)((v / res) % res), (int)(v % res)};
    int k = m.counts[v * 2];
    for (int a = 0; a < 3; ++a) {
      if (!((fl >> a) & 1)) continue;
      const float u0 = m.u[v], u1 = m.u[v + stride[a]];
      const float t = (m.thr - u0) / (u1 - u0);
      for (int c = 0; c < 3; ++c) {
        const float g = (float)xyz[c] + (c == a ? t : 0.0f);
        verts[(long)k * 3 + c] = g / ((float)res - 1.0f) * (bmax[c] - bmin[c]) + bmin[c];
      }
      ++k;
    }
    unsigned char dummy;
    const int idx = mc_cell(m.u, res, m.thr, xyz[0], xyz[1], xyz[2], &dummy);
    if (idx < 0) continue;
    const int t0 = m.counts[v * 2 + 1];
    for (int t = 0; t < kMcNumTris[idx]; ++t)
      for (int q = 0; q < 3; ++q) {
        const int ed = kMcTris[idx][t * 3 + q], a = ed >> 2, kk = ed & 3;
        const int o0 = a == 0 ? 1 : 0, o1 = a == 2 ? 1 : 2;
        const long vo = v + (kk & 1) * stride[o0] + (kk >> 1) * stride[o1];
        tris[(long)(t0 + t) * 3 + q] = m.counts[vo * 2] + mc_popcount3(m.flags[vo] & ((1u << a) - 1u));
      }
  }
}

void be_gen_rays(const GenRays& p, cnr_stream) {
  for (long i = 0; i < p.n; ++i) body_gen_rays(p, i);
}
void be_gen_rays_bwd(const GenRaysBwd& q, cnr_stream) {
  for (int cam = 0; cam < q.f.n_cams; ++cam) {
    double acc[14] = {0};   // (the HIP kernel sums in float in a fixed tree order; the tests compare both with autograd)
    for (long i = 0; i < q.f.n; ++i) {
      int c; float out[14];
      body_gen_rays_bwd1(q, i, &c, out);
      if (c == cam) for (int k = 0; k < 14; ++k) acc[k] += out[k];
    }
    for (int k = 0; k < 12; ++k) q.d_c2w[cam * 16 + k] = (float)acc[k];
    for (int k = 12; k < 16; ++k) q.d_c2w[cam * 16 + k] = 0.0f;
    q.d_focal_partial[cam * 2] = (float)acc[12]; q.d_focal_partial[cam * 2 + 1] = (float)acc[13];
  }
  for (int j = 0; j < 2; ++j) {
    float s = 0.0f;
    for (int c = 0; c < q.f.n_cams; ++c) s += q.d_focal_partial[c * 2 + j];
    q.d_focal[j] = s;
  }
}

void be_clip_adam(const AdamArgs& a, cnr_stream) {
  for (int k = 0; k < a.count; ++k) {
    const AdamTensor& t = a.t[k];
    float coef = 1.0f;
    if (a.max_norm > 0.0f) {
      float tot = 0.0f;   // per-chunk sums, then the chunks (the summation order differs from the HIP kernels': round-off level)
      for (long c0 = 0; c0 < t.n; c0 += kAdamChunk) {
        float ss = 0.0f;
        for (long i = c0; i < t.n && i < c0 + kAdamChunk; ++i) ss += t.g[i] * t.g[i];
        tot += ss;
      }
      coef = adam_clip_coef(tot, a.max_norm);
    }
    for (long i = 0; i < t.n; ++i) adam_update1(a, t.w + i, t.g[i] * coef, a.m + t.off + i, a.v + t.off + i);
  }
}

// the CPU emulation has no chain-fused kernels: the host orchestration falls back to the per-layer sequence
void be_pack_frags_many(const PackJob*, int, cnr_stream) {}
bool be_sdf_value_chain(const SdfValueChain&, cnr_stream) { return false; }
bool be_relu_chain_fwd(const ReluChainFwd&, cnr_stream) { return false; }
bool be_relu_chain_fwd_enabled() { return false; }
bool be_sdf_save_chain(const SdfSaveChain&, cnr_stream) { return false; }
bool be_sdf_grad_chain(const SdfGradChain&, cnr_stream) { return false; }
bool be_sweep0_ok(const LayerGemm&) { return false; }   // (nor the sweep launch that forms its layer's weight-gradient pair)
int be_sweep0_slots(long) { return 0; }
void be_sweep0_dw(const LayerGemm&, float*, int, cnr_stream) {}
bool be_narrow_bwd_ok(const NarrowBwd&) { return false; }   // (nor the one-pass backward of a narrow-input layer)
int be_narrow_bwd_slots(long) { return 0; }
void be_narrow_bwd(const NarrowBwd&, cnr_stream) {}

void be_upsample(const UpSample& p, cnr_stream) {
#pragma omp parallel for
  for (long ray = 0; ray < p.R; ++ray) {
    const int n = p.n, nsec = n - 1;
    const float* z = p.z + ray * p.ldz;
    const float* s = p.w_in ? nullptr : p.sdf + ray * p.lds;
    std::vector<float> rad(n), cs(n), w(n), cdf(n);
    float T = 1.0f, total = 0.0f;
    if (p.w_in) {
      for (int i = 0; i < nsec; ++i) { w[i] = p.w_in[ray * nsec + i] + 1e-5f; total += w[i]; }
    } else {
    for (int i = 0; i < n; ++i) {
      float x = p.o[ray * 3] + p.d[ray * 3] * z[i], y = p.o[ray * 3 + 1] + p.d[ray * 3 + 1] * z[i], q = p.o[ray * 3 + 2] + p.d[ray * 3 + 2] * z[i];
      rad[i] = sqrtf(x * x + y * y + q * q);
    }
    for (int i = 0; i < nsec; ++i) cs[i] = (s[i + 1] - s[i]) / (z[i + 1] - z[i] + 1e-5f);
    for (int i = 0; i < nsec; ++i) {
      float prev = i > 0 ? cs[i - 1] : 0.0f;
      float c = std::min(prev, cs[i]);
      c = std::min(std::max(c, -1e3f), 0.0f);
      if (!(rad[i] < 1.0f || rad[i + 1] < 1.0f)) c = c * 0.0f;
      float a = upsample_alpha(s[i], s[i + 1], z[i], z[i + 1], c, p.inv_s);
      w[i] = a * T + 1e-5f;
      T = T * (1.0f - a + 1e-7f);
      total += w[i];
    }
    }
    cdf[0] = 0.0f;
    float run = 0.0f;
    for (int i = 0; i < nsec; ++i) { run += w[i] / total; cdf[i + 1] = run; }
    for (int k = 0; k < p.m; ++k) {
      float u = p.u_in ? p.u_in[ray * p.m + k] : linspace_at(0.5f / (float)p.m, 1.0f - 0.5f / (float)p.m, p.m, k);
      int idx = (int)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
      int below = std::max(idx - 1, 0), above = std::min(idx, n - 1);
      float den = cdf[above] - cdf[below];
      if (den < 1e-5f) den = 1.0f;
      float t = (u - cdf[below]) / den;
      p.new_z[ray * p.m + k] = z[below] + t * (z[above] - z[below]);
    }
  }
}

void be_merge(const MergeZ& p, cnr_stream) {
#pragma omp parallel for
  for (long ray = 0; ray < p.R; ++ray) {
    std::vector<float> z(p.z + ray * p.ldz, p.z + ray * p.ldz + p.n), s(p.n, 0.0f);
    if (p.new_sdf) s.assign(p.sdf_in + ray * p.lds_in, p.sdf_in + ray * p.lds_in + p.n);
    const float* nz = p.new_z + ray * p.m;
    for (int i = 0; i < p.n; ++i) {
      int cnt = 0;
      for (int j = 0; j < p.m; ++j) cnt += nz[j] < z[i] ? 1 : 0;
      p.z[ray * p.ldz + i + cnt] = z[i];
      if (p.new_sdf) p.sdf_out[ray * p.lds_out + i + cnt] = s[i];
    }
    for (int j = 0; j < p.m; ++j) {
      int lo = (int)(std::upper_bound(z.begin(), z.end(), nz[j]) - z.begin());
      int rank = 0;
      for (int q = 0; q < p.m; ++q) rank += (nz[q] < nz[j] || (nz[q] == nz[j] && q < j)) ? 1 : 0;
      p.z[ray * p.ldz + lo + rank] = nz[j];
      if (p.new_sdf) p.sdf_out[ray * p.lds_out + lo + rank] = p.new_sdf[ray * p.m + j];
    }
  }
}

struct SampleF {
  float z, dist, relax, inside, gn, g[3];
  AlphaOut a;
};
static SampleF sample_fwd(const float* z, int j, int M, float sample_dist, const float* o, const float* d, const float* sdf,
                          const float* g, long pt, float inv_s, float r) {
  SampleF q;
  q.z = z[j];
  q.dist = j + 1 < M ? z[j + 1] - q.z : sample_dist;
  float mid = q.z + q.dist * 0.5f;
  float x = o[0] + d[0] * mid, y = o[1] + d[1] * mid, w = o[2] + d[2] * mid;
  float pn = sqrtf(x * x + y * y + w * w);
  q.inside = pn < 1.0f ? 1.0f : 0.0f;
  q.relax = pn < 1.2f ? 1.0f : 0.0f;
  for (int k = 0; k < 3; ++k) q.g[k] = g[pt * 3 + k];
  q.gn = sqrtf(q.g[0] * q.g[0] + q.g[1] * q.g[1] + q.g[2] * q.g[2]);
  q.a = alpha_forward(sdf[pt], q.g, d, q.dist, inv_s, r);
  return q;
}

void be_composite_fwd(const CompositeFwd& p, cnr_stream) {
  const float inv_s = std::min(std::max(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);
#pragma omp parallel for
  for (long ray = 0; ray < p.R; ++ray) {
    const int M = p.M;
    const float* z = p.z + ray * M;
    float T = 1.0f, wsum = 0.f, wmax = -1.f, dep = 0.f, col[3] = {0, 0, 0}, gcl[3] = {0, 0, 0}, e0 = 0.f, e1 = 0.f;
    for (int j = 0; j < M; ++j) {
      long pt = ray * M + j;
      SampleF q = sample_fwd(z, j, M, p.sample_dist, p.o + ray * 3, p.d + ray * 3, p.sdf, p.g, pt, inv_s, p.cos_anneal);
      float w = q.a.alpha * T;
      T = T * (1.0f - q.a.alpha + 1e-7f);
      wsum += w; wmax = std::max(wmax, w); dep += w * q.z;
      if (p.color) for (int k = 0; k < 3; ++k) col[k] += w * p.color[pt * p.ldcolor + k];
      if (p.gcolor) for (int k = 0; k < 3; ++k) gcl[k] += w * p.gcolor[pt * p.ldg + k];
      e0 += q.relax * (q.gn - 1.0f) * (q.gn - 1.0f); e1 += q.relax;
      p.weights[pt] = w; p.cdf_fine[pt] = q.a.pc; p.inside_sphere[pt] = q.inside;
      if (p.sdf_s) p.sdf_s[pt] = p.sdf[pt];
      if (p.color_s && p.color) for (int k = 0; k < 3; ++k) p.color_s[pt * 3 + k] = p.color[pt * p.ldcolor + k];
      if (p.gcolor_s && p.gcolor) for (int k = 0; k < 3; ++k) p.gcolor_s[pt * 3 + k] = p.gcolor[pt * p.ldg + k];
    }
    for (int k = 0; k < 3; ++k) {
      float cc = col[k];
      if (p.background_rgb) cc = cc + p.background_rgb[k] * (1.0f - wsum);
      p.color_fine[ray * 3 + k] = cc;
      if (p.global_color) p.global_color[ray * 3 + k] = gcl[k];
    }
    p.weight_sum[ray] = wsum; p.weight_max[ray] = wmax; p.depth[ray] = dep; p.s_val[ray] = 1.0f / inv_s;
    p.eik_partial[ray * 2] = e0; p.eik_partial[ray * 2 + 1] = e1;
    if (p.delta && p.delta_ray_sum) {
      float drs = 0.f;
      for (long i = ray * M * 3; i < (ray + 1) * M * 3; ++i) drs += p.delta[i];
      p.delta_ray_sum[ray] = drs;
    }
  }
}

void be_composite_bwd(const CompositeBwd& p, cnr_stream) {
  const float inv_s = std::min(std::max(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);
  const float dge = p.d_gradient_error ? p.d_gradient_error[0] : 0.0f;
  const float eik_den = p.eik_sums[1] + 1e-5f;
#pragma omp parallel for
  for (long ray = 0; ray < p.R; ++ray) {
    const int M = p.M;
    const float* z = p.z + ray * M;
    const float* d = p.d + ray * 3;
    std::vector<SampleF> q(M);
    std::vector<float> T(M), w(M), wbar(M), S(M);
    float t = 1.0f, wmax = -1.0f;
    int amax = 0;
    for (int j = 0; j < M; ++j) {
      q[j] = sample_fwd(z, j, M, p.sample_dist, p.o + ray * 3, d, p.sdf, p.g, ray * M + j, inv_s, p.cos_anneal);
      T[j] = t; w[j] = q[j].a.alpha * t; t = t * (1.0f - q[j].a.alpha + 1e-7f);
      if (w[j] > wmax) { wmax = w[j]; amax = j; }
    }
    float dcol[3] = {0, 0, 0}, dglob[3] = {0, 0, 0};
    if (p.d_color_fine) for (int k = 0; k < 3; ++k) dcol[k] = p.d_color_fine[ray * 3 + k];
    if (p.d_global_color) for (int k = 0; k < 3; ++k) dglob[k] = p.d_global_color[ray * 3 + k];
    float dws = p.d_weight_sum ? p.d_weight_sum[ray] : 0.0f;
    if (p.background_rgb) for (int k = 0; k < 3; ++k) dws -= dcol[k] * p.background_rgb[k];
    const float ddepth = p.d_depth ? p.d_depth[ray] : 0.0f, dwmax = p.d_weight_max ? p.d_weight_max[ray] : 0.0f;
    for (int j = 0; j < M; ++j) {
      long pt = ray * M + j;
      float wb = 0.0f;
      for (int k = 0; k < 3; ++k) wb += dcol[k] * p.color[pt * p.ldcolor + k];
      if (p.gcolor) for (int k = 0; k < 3; ++k) wb += dglob[k] * p.gcolor[pt * p.ldg + k];
      wb += dws + ddepth * q[j].z;
      if (p.d_weights) wb += p.d_weights[pt];
      if (j == amax) wb += dwmax;
      wbar[j] = wb;
    }
    float run = 0.0f;
    for (int j = M - 1; j >= 0; --j) { S[j] = run; run += wbar[j] * w[j]; }
    float dinvs = 0.0f, drd[3] = {0, 0, 0};
    for (int j = 0; j < M; ++j) {
      long pt = ray * M + j;
      float dalpha = wbar[j] * T[j] - S[j] / (1.0f - q[j].a.alpha + 1e-7f);
      AlphaGrad ag = alpha_backward(q[j].a, q[j].dist, inv_s, p.cos_anneal, dalpha, p.d_cdf ? p.d_cdf[pt] : 0.0f);
      dinvs += ag.d_inv_s;
      float ecoef = (q[j].relax > 0.0f && q[j].gn > 0.0f) ? dge / eik_den * 2.0f * (q[j].gn - 1.0f) / q[j].gn : 0.0f;
      for (int k = 0; k < 3; ++k) {
        float gb = ag.d_tc * d[k] + ecoef * q[j].g[k];
        if (p.d_gradients) gb += p.d_gradients[pt * 3 + k];
        p.gbar[pt * 4 + k] = gb;
        drd[k] += ag.d_tc * q[j].g[k];
      }
      p.gbar[pt * 4 + 3] = 0.0f;
      if (p.d_z) { p.d_z[pt * 2] = ddepth * w[j]; p.d_z[pt * 2 + 1] = ag.d_dist; }
      p.ztop[pt * p.ldztop + p.ztop_col] = (ag.d_sdf + (p.d_sdf_s ? p.d_sdf_s[pt] : 0.0f)) / p.sdf_scale;
          for (int k = p.ztop_col + 1; k < p.ldztop; ++k) p.ztop[pt * p.ldztop + k] = 0.0f;   // (the pad columns behind it: one launch less than zeroing them apart)
      for (int k = 0; k < 3; ++k) {
        float cbar = dcol[k] * w[j] + (p.d_color_s ? p.d_color_s[pt * 3 + k] : 0.0f);
        if (p.has_relight) {
          float relit = p.color[pt * p.ldcolor + k], gc = p.gcolor[pt * p.ldg + k];
          float tbar, gca = dglob[k] * w[j] + (p.d_gcolor_s ? p.d_gcolor_s[pt * 3 + k] : 0.0f);
          if (p.inv_sigmoid) {
            tbar = cbar * relit * (1.0f - relit);
            gca += tbar * inverse_sigmoid_grad(gc);
          } else {
            float pass = (relit > 0.0f && relit < 1.0f) ? 1.0f : 0.0f;
            float sg = relit - gc + 0.5f;
            tbar = cbar * pass * sg * (1.0f - sg);
            gca += cbar * pass;
          }
          p.dtop[pt * p.ldtop + k] = tbar + (p.d_delta_relight ? p.d_delta_relight[pt * 3 + k] : 0.0f) + (p.d_delta_relight_ray ? p.d_delta_relight_ray[ray] : 0.0f);
          p.gc_a[pt * p.ldtop + k] = gca;
        } else {
          p.gc_a[pt * p.ldtop + k] = cbar;
        }
      }
      if (p.has_relight) for (int k = 3; k < p.ldtop; ++k) p.dtop[pt * p.ldtop + k] = 0.0f;
      for (int k = 3; k < p.ldtop; ++k) p.gc_a[pt * p.ldtop + k] = 0.0f;
    }
    if (p.d_s_val) dinvs += -p.d_s_val[ray] / (inv_s * inv_s);
    p.dinvs_partial[ray] = dinvs;
    if (p.d_rays_d) for (int k = 0; k < 3; ++k) p.d_rays_d[ray * 3 + k] = drd[k];
  }
}

void be_rays_grad_finish(const RaysGradFinish& p, cnr_stream) {
  const int npe = p.multires_view > 0 ? 3 + 6 * p.multires_view : 3;
  for (long ray = 0; ray < p.R; ++ray) {
    float so[3] = {0, 0, 0}, sd[3] = {0, 0, 0}, spe[27];
    for (int q = 0; q < 27; ++q) spe[q] = 0.0f;
    const float* dr = p.d + ray * 3;
    float snear = 0.0f, sfar = 0.0f;
    std::vector<float> dmid(p.M);
    for (int j = 0; j < p.M; ++j) {
      long pt = ray * p.M + j;
      float z0 = p.z[pt];
      float dist = j + 1 < p.M ? p.z[pt + 1] - z0 : p.sample_dist;
      float mid = z0 + dist * 0.5f;
      dmid[j] = 0.0f;
      for (int k = 0; k < 3; ++k) { float pb = p.pbar[pt * 4 + k]; so[k] += pb; sd[k] += pb * mid; dmid[j] += pb * dr[k]; }
      for (int q = 0; q < npe; ++q) {
        if (p.daux_dir_c) spe[q] += p.daux_dir_c[pt * p.lddir + 6 + q];
        if (p.daux_dir_r) spe[q] += p.daux_dir_r[pt * p.lddir + 6 + q];
      }
    }
    if (p.dz_parts && p.d_near && p.d_far) {
      for (int j = 0; j < p.M; ++j) {
        long pt = ray * p.M + j;
        float dz = p.dz_parts[pt * 2] + dmid[j];
        if (j + 1 < p.M) dz -= p.dz_parts[pt * 2 + 1] + 0.5f * dmid[j];
        if (j > 0) dz += p.dz_parts[(pt - 1) * 2 + 1] + 0.5f * dmid[j - 1];
        float lin = linspace_at(0.0f, 1.0f, p.M, j);
        snear += dz * (1.0f - lin);
        sfar += dz * lin;
      }
      p.d_near[ray] = snear; p.d_far[ray] = sfar;
    }
    if (!(p.d_o && p.d_d)) continue;
    for (int k = 0; k < 3; ++k) {
      float dk = dr[k];
      float acc = sd[k] + p.d_rays_d_alpha[ray * 3 + k] + spe[k];
      float f = 1.0f;
      for (int m = 0; m < p.multires_view; ++m) {
        acc += f * (cosf(dk * f) * spe[3 + 6 * m + k] - sinf(dk * f) * spe[6 + 6 * m + k]);
        f *= 2.0f;
      }
      p.d_d[ray * 3 + k] = acc;
      p.d_o[ray * 3 + k] = so[k];
    }
  }
}

}  // namespace cnr

namespace cnr {
void be_prune_count(const PruneCount& p, cnr_stream) {
  for (long r = 0; r < p.R; ++r) {
    int c = 0;
    for (int j = 0; j < p.M; ++j) c += p.weights[r * p.M + j] >= p.eps ? 1 : 0;
    p.counts[r] = c;
  }
}
void be_prune_scan(const PruneScan& p, cnr_stream) {
  int run = 0;
  for (long r = 0; r < p.R; ++r) { p.offsets[r] = run; run += p.counts[r]; }
  p.offsets[p.R] = run;
}
void be_prune_gather(const PruneGather& p, cnr_stream) {
  for (long r = 0; r < p.R; ++r) {
    int k = p.offsets[r];
    for (int j = 0; j < p.M; ++j) {
      long pt = r * p.M + j;
      if (!(p.weights[pt] >= p.eps)) {
        if (p.zero_gcol) {
          for (int c = 0; c < 4; ++c) { p.zero_gcol[pt * 4 + c] = 0.f; if (p.zero_relit) p.zero_relit[pt * 4 + c] = 0.f; }
          if (p.zero_delta) for (int c = 0; c < 3; ++c) p.zero_delta[pt * 3 + c] = 0.f;
        }
        continue;
      }
      p.idx[k] = (int)pt;
      if (p.featx_c) {
        memcpy(p.featx_c + (long)k * p.ldfx, p.featx + pt * p.ldfx, sizeof(float) * p.ldfx);
        memcpy(p.aux_c + (long)k * kAux, p.aux + pt * kAux, sizeof(float) * kAux);
      }
      ++k;
    }
  }
}
void be_prune_scatter(const PruneScatter& p, cnr_stream) {
  for (long i = 0; i < *p.count; ++i) {
    long pt = p.idx[i];
    for (int c = 0; c < 4; ++c) p.gcol[pt * 4 + c] = p.gcol_c[i * 4 + c];
    if (p.relit) for (int c = 0; c < 4; ++c) p.relit[pt * 4 + c] = p.relit_c[i * 4 + c];
    if (p.delta) for (int c = 0; c < 3; ++c) p.delta[pt * 3 + c] = p.delta_c[i * 3 + c];
  }
}
}  // namespace cnr
