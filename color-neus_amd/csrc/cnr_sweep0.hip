// Second-order sweep launch of a NARROW-input layer (K <= 48: the first SDF layer, 39 embedding columns -> 256) together with the
// gradient-chain weight-gradient pair of the same layer, for gfx950 (MI355X / CDNA4).
//
// Why: the sweep launch of layer 0 reads z_0 and v_0 (2 KB per point) for its epilogue, and the weight-gradient GEMM of the pair
// (u_0 = sp'(z_0) v_0, cbar) used to read both again -- 1.07 GB and, as an FP32-MFMA launch, 0.27 ms of matrix time per step for a
// 256 x 39 result.  Here the pair is formed from the values the epilogue holds.
//
// Layout: one workgroup per point range, 8 waves x 32 output columns, weights of the 3 k16 blocks resident in registers (cnr_gemm_ws.h),
// 32-point tiles.  A wave OWNS its 32 rows of dW (dW[j][c] = sum_pt u[pt][j] cbar[pt][c], j = its output columns), so everything about the
// weight gradient is wave-private: after the epilogue of a tile the wave writes u', scaled and split hi / lo, row-major into its own LDS
// strip (one 8-byte store per plane and lane), and takes both MFMA fragments -- u' from the strip, cbar' from the row-major planes the product
// used -- by the LDS transpose read (ws_tr8, cnr_gemm_ws.h; round 6: 8 + 16 reads per tile and wave instead of 32 two-byte stores, 4 sixteen-byte
// reads and 64 two-byte reads), and issues 2 x 2 x 3 MFMAs into a [32 x 64] accumulator block.
// Scaling as in cnr_gemm_fdw.hip: cbar' = cbar * ss[pt] are the product's planes; u' = u * 2^G / ss[pt]; G = 1 + the running minimum of
// log2(ss * se), se = the scale that lifts the wave's 32 values of a row into the top f16 binade -- per WAVE here (no saved row scales of u_0
// exist, and none are needed: the rows of dW never mix waves); lowering G rescales the accumulators by the exact power of two.
// Fixed summation order per range: bitwise deterministic.
#include "cnr_gemm_ws.h"

namespace cnr {

constexpr int S0_NKB = 3;                          // k16 blocks (K <= 48)
constexpr int S0_ALD = S0_NKB * 32 + 16;           // bytes per LDS row of one cbar plane
constexpr int S0_APLANE = WS_TP * S0_ALD;
constexpr int S0_ABUF = 2 * S0_APLANE + 256;       // two planes + rs[32] (1 / row scale) + ss[32] (row scale)
constexpr int S0_YLD = 80;                         // bytes per POINT row of one u' plane (row-major [point][32 columns of the wave], round 6): 64 B + 16 -- 20 dwords = 4 banks mod 16 (ws_tr8)
constexpr int S0_YPLANE = 32 * S0_YLD;
constexpr int S0_YBUF = 2 * S0_YPLANE;             // per wave
constexpr int S0_OFF_T = 2 * S0_ABUF;
constexpr int S0_OFF_Y = S0_OFF_T + 8 * 32 * WS_TLD * 4;
constexpr int S0_LDS = S0_OFF_Y + 8 * S0_YBUF;
constexpr int S0_GBIG = 0x3f000000;

__device__ __forceinline__ float s0_yscale(float sx, int G) {   // 2^G / sx for a power-of-two sx > 0 (exponent arithmetic; underflow flushes to 0)
  const unsigned bits = __float_as_uint(sx);
  const int field = G - (int)((bits >> 23) & 0xff) + 254;
  return field < 1 ? 0.0f : __uint_as_float((unsigned)(field > 254 ? 254 : field) << 23);
}

__global__ __launch_bounds__(WS_THREADS, 1) void sweep0_dw_kernel(const LayerGemm g, float* partial, int ldk, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long Pn = g.P;
  const long ntiles = (Pn + WS_TP - 1) / WS_TP;
  const long t0 = (long)blockIdx.x * tiles_per_wg;
  long t1 = t0 + tiles_per_wg;
  if (t1 > ntiles) t1 = ntiles;
  const int n = t1 > t0 ? (int)(t1 - t0) : 0;
  const int c0 = wave * 32;
  const int kpad = ((g.K + 15) >> 4) * 16;

  f16x8 w1[S0_NKB], w2[S0_NKB];
  {
    const unsigned short* wp = g.Wp + (long)(c0 + (lane & 31)) * g.ldw + (lane >> 5) * 8;
#pragma unroll
    for (int kb = 0; kb < S0_NKB; ++kb) {
      const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
      w1[kb] = kb * 16 < kpad ? *reinterpret_cast<const f16x8*>(wp + kb * 16) : z8;
      w2[kb] = kb * 16 < kpad ? *reinterpret_cast<const f16x8*>(wp + g.wp_stride + kb * 16) : z8;
    }
  }
  const int ecol = c0 + (lane & 7) * 4;
  const f4 wsc = *reinterpret_cast<const f4*>(g.wscale + ecol);
  float* T = reinterpret_cast<float*>(smem_s + S0_OFF_T) + wave * (32 * WS_TLD);
  unsigned char* Yw = smem_s + S0_OFF_Y + wave * S0_YBUF;

  // staging map: 16 threads per row, 4 columns each (K <= 48: 12 of them hold data)
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  const bool pv = scol < kpad;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  f4 sraw = z4;
  auto s_fetch = [&](int i) {
    long row = (t0 + (i < n ? i : (n > 0 ? n - 1 : 0))) * WS_TP + srow;
    if (row >= Pn) row = Pn - 1;
    if (pv) sraw = *reinterpret_cast<const f4*>(g.A.a + row * g.A.lda + scol);
  };
  auto s_put = [&](int i, int buf) {
    const long prow = (t0 + i) * WS_TP + srow;
    const f4 v = (pv && prow < Pn) ? sraw : z4;
    float mx = ws_absmax4(v);
    mx = cnr_max16(mx);
    const bool valid = mx > 0.0f && mx < 3.0e38f;
    float sc = 1.0f;
    if (valid) { int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100; sc = ldexpf(1.0f, 14 - e_); }
    unsigned char* B = smem_s + buf * S0_ABUF;
    // every column the product loop reads (S0_NKB k16 blocks) is written: columns in [kpad, 48) of a K <= 32 layer get zeros, not whatever
    // an earlier kernel left in LDS (NaN / Inf bit patterns times the zero weights of those blocks would poison the accumulators)
    if (scol < S0_NKB * 16) ws_put4(v, sc, B + srow * S0_ALD + scol * 2, S0_APLANE);
    if ((tid & 15) == 0) {
      float* rs = reinterpret_cast<float*>(B + 2 * S0_APLANE);
      const float ssv = valid ? sc : (mx == 0.0f ? 0.0f : __builtin_nanf(""));
      rs[srow] = cnr_pow2_rcp(sc);
      rs[32 + srow] = ssv;
      if (g.rs_out && prow < Pn) g.rs_out[prow] = ssv;
    }
  };
  // side inputs of the epilogue: z and v of this lane's 4 columns in 4 row groups
  EpiRaw4 er[4];
  auto e_fetch = [&](int i) {
    const long t = t0 + (i < n ? i : (n > 0 ? n - 1 : 0));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      long row = t * WS_TP + (lane >> 3) + 8 * q;
      if (row >= Pn) row = Pn - 1;
      er[q] = epi_fetch4_plain<EK_SWEEP>(g.E, row, ecol);
    }
  };

  f32x16 dacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[i][r] = 0.0f;
  int G = S0_GBIG;

  if (n > 0) {
    s_fetch(0);
    s_put(0, 0);
    s_fetch(1);
    e_fetch(0);
  }
  cnr_lds_barrier();
  for (int i = 0; i < n; ++i) {
    const int buf = i & 1;
    const long t = t0 + i;
    if (i + 1 < n) s_put(i + 1, buf ^ 1);
    s_fetch(i + 2);
    const unsigned char* B = smem_s + buf * S0_ABUF;
    // ---- product of this wave's 32 columns
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    {
      const unsigned char* Ab = B + (lane & 31) * S0_ALD + (lane >> 5) * 16;
#pragma unroll
      for (int kb = 0; kb < S0_NKB; ++kb) {
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32);
        const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + S0_APLANE + kb * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0);
      }
    }
    const float* rs = reinterpret_cast<const float*>(B + 2 * S0_APLANE);
    const float* ssr = rs + 32;
    {
      const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * WS_TLD + cl] = acc[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- epilogue (same arithmetic as epi_finish4_plain<EK_SWEEP>) + the pair's n-side operand u = sp'(z) v
    f4 u[4];
    int qmin = S0_GBIG;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = (lane >> 3) + 8 * q, cc = (lane & 7) * 4;
      const long row = t * WS_TP + rr;
      const float rsc = rs[rr];
      f4 v = *reinterpret_cast<const f4*>(T + rr * WS_TLD + cc);
      v.x *= rsc * wsc.x; v.y *= rsc * wsc.y; v.z *= rsc * wsc.z; v.w *= rsc * wsc.w;
      const f4 zz = er[q].a, vv = er[q].b;
      const float s1x = softplus100_d1(zz.x), s1y = softplus100_d1(zz.y), s1z = softplus100_d1(zz.z), s1w = softplus100_d1(zz.w);
      const float vx = vv.x * g.E.vscale, vy = vv.y * g.E.vscale, vz = vv.z * g.E.vscale, vw = vv.w * g.E.vscale;
      f4 o1, o2;
      o1.x = softplus100_d2(zz.x) * vx * v.x; o2.x = s1x * v.x;
      o1.y = softplus100_d2(zz.y) * vy * v.y; o2.y = s1y * v.y;
      o1.z = softplus100_d2(zz.z) * vz * v.z; o2.z = s1z * v.z;
      o1.w = softplus100_d2(zz.w) * vw * v.w; o2.w = s1w * v.w;
      const bool live = row < Pn;
      if (live) {
#if WS_NT_SWEEP
        __builtin_nontemporal_store(o1, reinterpret_cast<f4*>(g.E.o1 + row * g.E.ld1 + ecol));
#else
        *reinterpret_cast<f4*>(g.E.o1 + row * g.E.ld1 + ecol) = o1;
#endif
        *reinterpret_cast<f4*>(g.E.o2 + row * g.E.ld2 + ecol) = o2;
      }
      u[q].x = live ? s1x * vx : 0.0f; u[q].y = live ? s1y * vy : 0.0f; u[q].z = live ? s1z * vz : 0.0f; u[q].w = live ? s1w * vw : 0.0f;
      float mx = ws_absmax4(u[q]);
      mx = cnr_max8(mx);   // (the 8 lanes of a row piece: DPP moves, not LDS-crossbar shuffles)
      const float ss = ssr[rr];
      if (mx > 0.0f && mx < 3.0e38f && ss > 0.0f) {
        int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100;
        const int e = (int)((__float_as_uint(ss) >> 23) & 0xff) - 127 + 14 - e_;   // log2(ss * se)
        qmin = e < qmin ? e : qmin;
      }
    }
    e_fetch(i + 1);
    qmin = cnr_pair32_min(cnr_pair16_min(cnr_ror8_min(qmin)));
    qmin = __builtin_amdgcn_readfirstlane(qmin);
    if (qmin < S0_GBIG && qmin + 1 < G) {
      if (G < S0_GBIG) {
        const int dlt = qmin + 1 - G;
        const float u1 = ldexpf(1.0f, dlt / 2), u2 = ldexpf(1.0f, dlt - dlt / 2);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) dacc[b][r] = dacc[b][r] * u1 * u2;
      }
      G = qmin + 1;
    }
    // ---- u' = u * 2^G / ss, split hi / lo, row-major ([point][column], one 8-byte store per plane) into this wave's strip
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = (lane >> 3) + 8 * q, cc = (lane & 7) * 4;
      const float ss = ssr[rr];
      const float ys = ss > 0.0f ? s0_yscale(ss, G) : ss;   // (0 for an all-zero cbar row, NaN for a non-finite one)
      f4 e = u[q];
      e.x *= ys; e.y *= ys; e.z *= ys; e.w *= ys;
      unsigned char* yrow = Yw + rr * S0_YLD + cc * 2;
      ws_f16x4 h1, h2;
      h1[0] = (_Float16)e.x; h1[1] = (_Float16)e.y; h1[2] = (_Float16)e.z; h1[3] = (_Float16)e.w;
      h2[0] = (_Float16)(e.x - (float)h1[0]); h2[1] = (_Float16)(e.y - (float)h1[1]);
      h2[2] = (_Float16)(e.z - (float)h1[2]); h2[3] = (_Float16)(e.w - (float)h1[3]);
      *reinterpret_cast<ws_f16x4*>(yrow) = h1;
      *reinterpret_cast<ws_f16x4*>(yrow + S0_YPLANE) = h2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- dW[j][c] += sum over the tile's points of u'[pt][j] * cbar'[pt][c]
    {
      const int m = lane & 31, kg = lane >> 5;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        // both fragments out of row-major planes by the LDS transpose read (ws_tr8, cnr_gemm_ws.h: block rows 4 points apart; the cbar planes'
        // row stride of 112 B puts them 48 banks apart, as good as 16)
        const int prow = kb * 16 + kg * 2 + 4 * ((lane & 15) >> 2), pcol = (m & 16) + (lane & 3) * 4;
        const unsigned char* ysrc = Yw + prow * S0_YLD + pcol * 2;
        const f16x8 a1 = ws_tr8(ysrc, S0_YLD);
        const f16x8 a2 = ws_tr8(ysrc + S0_YPLANE, S0_YLD);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int c = cb * 32 + m;
          const unsigned char* src = B + prow * S0_ALD + (cb * 32 + pcol) * 2;   // (columns >= kpad: whatever lies behind the row, masked below)
          f16x8 b1 = ws_tr8(src, S0_ALD);
          f16x8 b2 = ws_tr8(src + S0_APLANE, S0_ALD);
          if (c >= kpad) {
            const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            b1 = z8; b2 = z8;
          }
          f32x16 cacc = dacc[cb];
          cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, cacc, 0, 0, 0);
          cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, cacc, 0, 0, 0);
          cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, cacc, 0, 0, 0);
          dacc[cb] = cacc;
        }
      }
    }
    cnr_lds_barrier();
  }
  // ---- partial sums of this range: rows c0 .. c0 + 31 of [Npad = 256][ldk], every slot written in full
  if (G >= S0_GBIG) G = 0;
  const float u1 = ldexpf(1.0f, -(G / 2)), u2 = ldexpf(1.0f, -(G - G / 2));
  float* out = partial + (long)blockIdx.x * 256 * ldk;
  const int m = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int c = cb * 32 + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = c0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (c < ldk) out[(long)j * ldk + c] = c < kpad ? dacc[cb][r] * u1 * u2 : 0.0f;
    }
  }
}

// workgroups (= partial-sum slots) of the launch for P points: one per CU as soon as there are enough tiles
int be_sweep0_slots(long P) {
  const long ntiles = (P + WS_TP - 1) / WS_TP;
  if (ntiles <= 0) return 0;
  const long tpw = (ntiles + 255) / 256;
  return (int)((ntiles + tpw - 1) / tpw);
}

bool be_sweep0_ok(const LayerGemm& g) {
  const bool off = debug_flags().no_sweep0 || debug_flags().no_fdw;   // debugging aids: separate launches
  const Epi& e = g.E;
  return !off && g.A.kind == VK_DIRECT && g.A.scale == 1.0f && (g.A.lda & 3) == 0 && g.K >= 1 && g.K <= 48 && g.A.lda >= ((g.K + 15) / 16) * 16 && g.N == 256 &&
         g.col0 == 0 && g.first_col == 0 && g.P_dev == nullptr && g.P > 0 && g.Wp != nullptr && g.wscale != nullptr && g.ldw >= ((g.K + 15) / 16) * 16 && g.k_extra == 0 &&
         e.kind == EK_SWEEP && e.tail_src == nullptr && e.split == (1 << 30) && e.n_out == 256 && e.z && e.v && e.o1 && e.o2 &&
         (e.ldz & 3) == 0 && (e.ldv & 3) == 0 && (e.ld1 & 3) == 0 && (e.ld2 & 3) == 0 && g.dot_w == nullptr;
}

// sweep launch g + the pair's weight gradient into be_sweep0_slots(g.P) slots of [256][ldk] floats at `partial`
void be_sweep0_dw(const LayerGemm& g, float* partial, int ldk, cnr_stream s) {
  const long ntiles = (g.P + WS_TP - 1) / WS_TP;
  const long tpw = (ntiles + 255) / 256;
  const int grid = be_sweep0_slots(g.P);
  static DeviceOnce attr_once;
  if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep0_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  TimingScope ts_("sweep0_dw", 0, 300, g.P, g.N, g.K, 2, s, layer_gemm_bytes(g) + 4.0 * grid * 256.0 * ldk);
  hipLaunchKernelGGL(sweep0_dw_kernel, dim3(grid), dim3(WS_THREADS), S0_LDS, s, g, partial, ldk, (int)tpw);
  CNR_LAUNCH_CHECK("sweep0_dw");
}

}  // namespace cnr
