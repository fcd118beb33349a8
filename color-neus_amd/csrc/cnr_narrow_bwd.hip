// Backward of a NARROW-input layer (K <= 48 input columns, 256 outputs) in one pass over its output cotangent, for gfx950 (MI355X / CDNA4):
//     dX[pt][c] = sum_j X[pt][j] W[j][c]          (optional: the cotangent of the layer's 33 / 39 input columns)
//     dW[j][c]  = sum_pt X[pt][j] Y[pt][c]        (X: the 256-wide cotangent of the layer's outputs, Y: the layer's forward input)
//     db[j]     = sum_pt X[pt][j]                 (optional)
//
// Why: the relight in_layer and the first SDF layer used to cost a narrow FP32-MFMA layer launch plus a FP32-MFMA weight-gradient launch, each
// streaming the same 1 KB-per-point cotangent for a 48-column product.  Here the cotangent is read once and both products are split-f16 MFMAs.
//
// Layout: one workgroup per point range (= partial-sum slot), 8 waves, 32-point tiles, double buffered.  A tile of X is staged like a layer
// input (cnr_gemm_ws.h: exact power-of-two row scale ss, f16 hi / lo planes, row-major); the tile of Y gets its own row scale sy and is stored
// row-major as well (one 8-byte store per plane and thread) with a column of ones behind its ky columns: that column of dW is the bias gradient.  dX: waves 0 / 1 take the
// 2 x 32 columns; the f16 planes of W^T (48 rows) sit in LDS for the whole launch (K = 256: 16 k16 blocks x 3 MFMAs per tile and wave).
// dW: wave w owns rows j in [32 w, 32 w + 32): both fragments come out of the row-major planes by the LDS transpose read (ws_tr8, cnr_gemm_ws.h;
// round 6: 8 + 8 reads per tile and wave instead of 32 two-byte reads + 8 sixteen-byte reads of a transposed Y tile written with two-byte stores).  The planes carry ss[pt] sy[pt] X Y; every A element is multiplied by the exact power of two 2^(Gt - e[pt]) <= 1,
// e = log2(ss sy), Gt = min of e over the tile (an f16 multiply by a power of two down to the subnormal 2^-24: the product keeps the f16 subnormal
// grid, i.e. an absolute error of 2^-25 per element like any other split operand), so a tile's MFMAs form 2^Gt sum X Y; the tile result is folded
// into the fp32 totals with 2^-Gt.  Two staging register sets keep two tiles in flight; fixed order: bitwise deterministic.
// The same kernel without the weight gradient (DX only, optionally with X = sp'(X) * Xb) ends the forward gradient chain.
#include "cnr_gemm_ws.h"

namespace cnr {

constexpr int NB_ALD = 256 * 2 + 16;
constexpr int NB_APLANE = WS_TP * NB_ALD;
constexpr int NB_ABUF = 2 * NB_APLANE + 256;      // two planes + rs[32] (1 / ss) + e[32] (int: log2(ss sy); NB_EBIG: the row contributes nothing)
constexpr int NB_YCOLS = 48;
constexpr int NB_YLD = 144;                       // bytes per POINT row of one Y plane (row-major [point][column], round 6): 48 columns x 2 B + 48 -- 36 dwords = 4 banks mod 16 (ws_tr8)
constexpr int NB_YPLANE = WS_TP * NB_YLD;
constexpr int NB_YBUF = 2 * NB_YPLANE;
constexpr int NB_OFF_Y = 2 * NB_ABUF;
constexpr int NB_OFF_T = NB_OFF_Y + 2 * NB_YBUF;
constexpr int NB_OFF_F = NB_OFF_T + 2 * 32 * WS_TLD * 4;   // per wave: the 32 point factors of a tile as f16
constexpr int NB_OFF_W = NB_OFF_F + 8 * 64;
constexpr int NB_WPLANE = NB_YCOLS * NB_ALD;      // W^T planes: 48 rows (input columns) x 256
constexpr int NB_LDS_NODX = NB_OFF_W;
constexpr int NB_LDS_DX = NB_OFF_W + 2 * NB_WPLANE;
constexpr int NB_EBIG = 0x3f000000;
static_assert(NB_LDS_DX <= 160 * 1024, "LDS budget of one CU");

// DX: the input-side product (needs Wp); DW: the weight / bias gradient (needs Y, partial); SIG: X is the view softplus100'(X) * Xb (the end of the
// forward gradient chain: u_0 = sp'(z_0) v_0 -> the cotangent of the embedding, a DX-only launch)
template <bool DX, bool DW, bool SIG>
__global__ __launch_bounds__(WS_THREADS, 1) void narrow_bwd_kernel(const NarrowBwd p, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_n[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long Pn = p.P;
  const long ntiles = (Pn + WS_TP - 1) / WS_TP;
  const long t0 = (long)blockIdx.x * tiles_per_wg;
  long t1 = t0 + tiles_per_wg;
  if (t1 > ntiles) t1 = ntiles;
  const int n = t1 > t0 ? (int)(t1 - t0) : 0;
  const int c0 = wave * 32;
  const bool has_w = DX && c0 < 64;
  // ---- W^T planes of the dX product into LDS (rows = input columns c < 48; a row beyond w_rows is zero)
  f4 wsc = {1.f, 1.f, 1.f, 1.f};
  if constexpr (DX) {
    unsigned char* Wl = smem_n + NB_OFF_W;
    for (int idx = tid; idx < 2 * NB_YCOLS * 32; idx += WS_THREADS) {   // 16-byte pieces: [plane][row][32 pieces]
      const int pl = idx / (NB_YCOLS * 32), rem = idx - pl * (NB_YCOLS * 32), row = rem >> 5, pc = rem & 31;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row < p.w_rows) v = *reinterpret_cast<const f16x8*>(p.Wp + pl * p.wp_stride + (long)row * p.ldw + pc * 8);
      *reinterpret_cast<f16x8*>(Wl + pl * NB_WPLANE + row * NB_ALD + pc * 16) = v;
    }
    if (has_w) {
      const int col = c0 + (lane & 7) * 4;
      wsc.x = col + 0 < p.w_rows ? p.wscale[col + 0] : 1.0f; wsc.y = col + 1 < p.w_rows ? p.wscale[col + 1] : 1.0f;
      wsc.z = col + 2 < p.w_rows ? p.wscale[col + 2] : 1.0f; wsc.w = col + 3 < p.w_rows ? p.wscale[col + 3] : 1.0f;
    }
  }
  float* T = reinterpret_cast<float*>(smem_n + NB_OFF_T) + (wave & 1) * (32 * WS_TLD);

  // ---- staging map: 16 threads per row; X: 4 passes of 64 columns; Y: 4 columns per thread (12 threads of a row)
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  const bool ylive = scol < NB_YCOLS;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  // two staging register sets: tile j travels in set j & 1, so that two tiles are in flight while a third is being used (one tile in flight
  // left the launch latency-bound: 2.5 - 3.2 TB/s)
  struct RawTile { f4 x[4]; f4 xb[SIG ? 4 : 1]; f4 y; };
  RawTile ra, rb;
#pragma unroll
  for (int q = 0; q < 4; ++q) { ra.x[q] = z4; rb.x[q] = z4; }
#pragma unroll
  for (int q = 0; q < (SIG ? 4 : 1); ++q) { ra.xb[q] = z4; rb.xb[q] = z4; }
  ra.y = z4; rb.y = z4;
  const bool want_cs = p.colsum != nullptr;   // (then column ky of Y is a column of ones: ky < 48)
  auto s_fetch = [&](int i, RawTile& rt) __attribute__((always_inline)) {
    long row = (t0 + (i < n ? i : (n > 0 ? n - 1 : 0))) * WS_TP + srow;
    if (row >= Pn) row = Pn - 1;
    const float* src = p.X + row * p.ldx + scol;
#pragma unroll
    for (int q = 0; q < 4; ++q) rt.x[q] = *reinterpret_cast<const f4*>(src + 64 * q);
    if constexpr (SIG) {
      const float* srcb = p.Xb + row * p.ldxb + scol;
#pragma unroll
      for (int q = 0; q < 4; ++q) rt.xb[q] = *reinterpret_cast<const f4*>(srcb + 64 * q);
    }
    if constexpr (DW) rt.y = *reinterpret_cast<const f4*>(p.Y + row * p.ldy + (ylive ? scol : 0));   // (every lane loads: a branch here costs the compiler its count of loads in flight)
  };
  auto s_put = [&](int i, int buf, const RawTile& rt) __attribute__((always_inline)) {
    const bool live = (t0 + i) * WS_TP + srow < Pn;
    f4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 x = rt.x[q];
      if constexpr (SIG) {
        x.x = softplus100_d1(x.x) * rt.xb[q].x; x.y = softplus100_d1(x.y) * rt.xb[q].y;
        x.z = softplus100_d1(x.z) * rt.xb[q].z; x.w = softplus100_d1(x.w) * rt.xb[q].w;
      }
      v[q] = live ? x : z4;
    }
    f4 y = rt.y;
    y.x = (live && scol + 0 < p.ky) ? y.x : 0.0f; y.y = (live && scol + 1 < p.ky) ? y.y : 0.0f;
    y.z = (live && scol + 2 < p.ky) ? y.z : 0.0f; y.w = (live && scol + 3 < p.ky) ? y.w : 0.0f;
    if (want_cs && live) {   // the ones column
      if (scol + 0 == p.ky) y.x = 1.0f;
      if (scol + 1 == p.ky) y.y = 1.0f;
      if (scol + 2 == p.ky) y.z = 1.0f;
      if (scol + 3 == p.ky) y.w = 1.0f;
    }
    if (!ylive) y = z4;
    float mx = fmaxf(fmaxf(ws_absmax4(v[0]), ws_absmax4(v[1])), fmaxf(ws_absmax4(v[2]), ws_absmax4(v[3])));
    float my = ws_absmax4(y);
    mx = cnr_max16(mx); my = cnr_max16(my);
    const bool vx = mx > 0.0f && mx < 3.0e38f, vy = my > 0.0f && my < 3.0e38f;
    float sx = 1.0f, sy = 1.0f;
    if (vx) { int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100; sx = ldexpf(1.0f, 14 - e_); }
    if (vy) { int e_; (void)frexpf(my, &e_); if (e_ < -100) e_ = -100; sy = ldexpf(1.0f, 14 - e_); }
    unsigned char* B = smem_n + buf * NB_ABUF;
    unsigned char* dst = B + srow * NB_ALD + scol * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) ws_put4(v[q], sx, dst + 128 * q, NB_APLANE);
    if (DW && ylive) {
      unsigned char* yb = smem_n + NB_OFF_Y + buf * NB_YBUF + srow * NB_YLD + scol * 2;
      const float ya[4] = {y.x * sy, y.y * sy, y.z * sy, y.w * sy};
      ws_f16x4 h1, h2;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h1[j] = (_Float16)ya[j]; h2[j] = (_Float16)(ya[j] - (float)h1[j]); }
      *reinterpret_cast<ws_f16x4*>(yb) = h1;
      *reinterpret_cast<ws_f16x4*>(yb + NB_YPLANE) = h2;
    }
    if ((tid & 15) == 0) {
      float* rs = reinterpret_cast<float*>(B + 2 * NB_APLANE);
      rs[srow] = cnr_pow2_rcp(sx);
      // a row with a zero operand contributes nothing (its planes are zero); a non-finite row keeps factor 1 so that it poisons the sums
      const bool nonfin = !(mx < 3.0e38f) || !(my < 3.0e38f);
      int e = NB_EBIG;
      if (vx && vy) e = (int)((__float_as_uint(sx) >> 23) & 0xff) + (int)((__float_as_uint(sy) >> 23) & 0xff) - 254;
      if (DW) reinterpret_cast<int*>(rs)[32 + srow] = nonfin ? -NB_EBIG : e;
    }
  };

  f32x16 tot[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[b][r] = 0.0f;

  if (n > 0) {
    s_fetch(0, ra);
    s_put(0, 0, ra);
    s_fetch(1, rb);
    s_fetch(2, ra);
  }
  cnr_lds_barrier();
  const int m = lane & 31, kg = lane >> 5;
  auto tile = [&](const int i, const int buf) __attribute__((always_inline)) {
    const long t = t0 + i;
    const unsigned char* B = smem_n + buf * NB_ABUF;
    const float* rs = reinterpret_cast<const float*>(B + 2 * NB_APLANE);
    const int* er = reinterpret_cast<const int*>(rs) + 32;
    // ---- dX of this tile (waves 0 / 1)
    if (has_w) {
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
      const unsigned char* Ab = B + m * NB_ALD + kg * 16;
      const int wrow = c0 + m;
      const unsigned char* Wb = smem_n + NB_OFF_W + (wrow < NB_YCOLS ? wrow : 0) * NB_ALD + kg * 16;
      const bool wlive = wrow < NB_YCOLS;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32);
        const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + NB_APLANE + kb * 32);
        f16x8 w1 = *reinterpret_cast<const f16x8*>(Wb + kb * 32);
        f16x8 w2 = *reinterpret_cast<const f16x8*>(Wb + NB_WPLANE + kb * 32);
        if (!wlive) { w1 = z8; w2 = z8; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * kg) * WS_TLD + m] = acc[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = (lane >> 3) + 8 * q, cc = (lane & 7) * 4;
        const long row = t * WS_TP + rr;
        const float rsc = rs[rr];
        f4 v = *reinterpret_cast<const f4*>(T + rr * WS_TLD + cc);
        v.x *= rsc * wsc.x; v.y *= rsc * wsc.y; v.z *= rsc * wsc.z; v.w *= rsc * wsc.w;
        const int col = c0 + cc;
        if (row < Pn) {
          float* o = p.dx + row * p.lddx + col;
          if (col + 3 < p.ndx) *reinterpret_cast<f4*>(o) = v;
          else {
            if (col + 0 < p.ndx) o[0] = v.x;
            if (col + 1 < p.ndx) o[1] = v.y;
            if (col + 2 < p.ndx) o[2] = v.z;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (!DW) return;
    // ---- dW of this tile: rows c0 .. c0 + 31
    const int e_l = er[m];
    int Gt = e_l == -NB_EBIG ? NB_EBIG - 1 : e_l;        // (non-finite rows do not steer the exponent ...)
    Gt = cnr_pair16_min(cnr_min16(Gt));   // (min over the 32 lanes of a half: DPP row moves + one row swap)
    Gt = __builtin_amdgcn_readfirstlane(Gt);
    if (Gt == NB_EBIG - 1) Gt = 0;                       // (... but a tile that has nothing else still has to carry them into the sums)
    if (Gt < NB_EBIG) {
      f32x16 tacc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tacc[b][r] = 0.0f;
      const unsigned char* Yb = smem_n + NB_OFF_Y + buf * NB_YBUF;
      // factors 2^(Gt - e[pt]) as f16, one point per lane, exchanged through the wave's LDS strip: 0 when the row contributes nothing, 1 for a
      // non-finite row, subnormal powers of two down to 2^-24 (the product then keeps the f16 subnormal grid: absolute error 2^-25 like any other element)
      unsigned short* fb = reinterpret_cast<unsigned short*>(smem_n + NB_OFF_F + wave * 64);
      {
        const int d = Gt - e_l;                          // <= 0 for live rows
        unsigned short bits = (e_l == NB_EBIG || d < -24) ? (unsigned short)0 : d < -14 ? (unsigned short)(1 << (d + 24)) : (unsigned short)((15 + d) << 10);
        if (e_l == -NB_EBIG) bits = (unsigned short)(15 << 10);
        if (lane < 32) fb[ws_kslot(lane)] = bits;   // (in the k order of the transpose-read fragments)
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const f16x8 fv = *reinterpret_cast<const f16x8*>(fb + kb * 16 + kg * 8);
        const int prow = kb * 16 + kg * 2 + 4 * ((lane & 15) >> 2), pcol = (m & 16) + (lane & 3) * 4;   // this lane's piece of a [4 points][16 columns] block
        f16x8 a1 = ws_tr8(B + prow * NB_ALD + (c0 + pcol) * 2, NB_ALD);
        f16x8 a2 = ws_tr8(B + NB_APLANE + prow * NB_ALD + (c0 + pcol) * 2, NB_ALD);
        a1 = a1 * fv; a2 = a2 * fv;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int c = cb * 32 + m;
          const unsigned char* ysrc = Yb + prow * NB_YLD + (cb * 32 + pcol) * 2;   // (columns 48 .. 63 of the second block: the pad behind a row, masked below)
          f16x8 b1 = ws_tr8(ysrc, NB_YLD);
          f16x8 b2 = ws_tr8(ysrc + NB_YPLANE, NB_YLD);
          if (c >= NB_YCOLS) {
            const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            b1 = z8; b2 = z8;
          }
          f32x16 cacc = tacc[cb];
          cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, cacc, 0, 0, 0);
          cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, cacc, 0, 0, 0);
          cacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, cacc, 0, 0, 0);
          tacc[cb] = cacc;
        }
      }
      const float u1 = ldexpf(1.0f, -(Gt / 2)), u2 = ldexpf(1.0f, -(Gt - Gt / 2));
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[b][r] = fmaf(tacc[b][r] * u1, u2, tot[b][r]);
    }
  };
  // pairs of tiles: every path through the loop body issues the same loads in the same order, so the compiler's count of what is in flight
  // stays exact (a skipped half would make it wait for everything at the loop head); a last odd tile follows the loop
  for (int i = 0; i + 1 < n; i += 2) {
    s_put(i + 1, 1, rb);
    s_fetch(i + 3, rb);
    tile(i, 0);
    cnr_lds_barrier();
    if (i + 2 < n) s_put(i + 2, 0, ra);
    s_fetch(i + 4, ra);
    tile(i + 1, 1);
    cnr_lds_barrier();
  }
  if (n & 1) tile(n - 1, 0);
  if constexpr (!DW) return;
  // ---- partial sums of this range: [256][ldk], every slot written in full
  float* out = p.partial + (long)blockIdx.x * 256 * p.ldk;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int c = cb * 32 + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = c0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      if (c < p.ldk) out[(long)j * p.ldk + c] = c < p.ky ? tot[cb][r] : 0.0f;
      if (want_cs && c == p.ky) p.colsum[(long)blockIdx.x * 256 + j] = tot[cb][r];   // the ones column: bias gradient
    }
  }
}

int be_narrow_bwd_slots(long P) {
  const long ntiles = (P + WS_TP - 1) / WS_TP;
  if (ntiles <= 0) return 0;
  const long tpw = (ntiles + 255) / 256;
  return (int)((ntiles + tpw - 1) / tpw);
}

bool be_narrow_bwd_ok(const NarrowBwd& p) {
  // debugging aid: narrow layer launch + weight-gradient launch as before.  Only the BACKWARD forms (those with a weight gradient): the
  // product-only launch that ends the forward gradient chain has its own switch below, so that this one leaves the forward pass untouched
  const bool off_bwd = debug_flags().no_narrow_bwd;
  const bool off = off_bwd && p.partial != nullptr;
  const bool dx_ok = p.Wp == nullptr || (p.wscale && p.dx && p.ldw == 256 && p.w_rows >= 1 && p.ndx >= 1 && p.ndx <= NB_YCOLS && p.ndx <= p.w_rows && (p.lddx & 3) == 0 && p.lddx >= ((p.ndx + 3) & ~3));
  const bool dw_ok = p.partial == nullptr ? (p.Wp != nullptr && p.colsum == nullptr)
                                          : (p.Y && (p.ldy & 3) == 0 && p.ky >= 1 && p.ky <= NB_YCOLS && p.ldy >= NB_YCOLS && p.ldk >= p.ky && p.ldk <= 64 &&
                                             (p.colsum == nullptr || p.ky < NB_YCOLS));
  const bool off_dx = debug_flags().no_narrow_dx;   // debugging aid: the FP32-MFMA layer kernel for the product-only launches
  const bool sig_ok = p.Xb == nullptr || ((p.ldxb & 3) == 0 && p.ldxb >= 256 && p.partial == nullptr);   // (the view form: DX-only launches)
  if (p.partial == nullptr && off_dx) return false;
  return !off && p.P > 0 && p.X && (p.ldx & 3) == 0 && p.ldx >= 256 && dx_ok && dw_ok && sig_ok;
}

void be_narrow_bwd(const NarrowBwd& p, cnr_stream s) {
  const long ntiles = (p.P + WS_TP - 1) / WS_TP;
  const long tpw = (ntiles + 255) / 256;
  const int grid = be_narrow_bwd_slots(p.P);
  static DeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&narrow_bwd_kernel<true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&narrow_bwd_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&narrow_bwd_kernel<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&narrow_bwd_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const bool dw = p.partial != nullptr;
  const double bytes = 4.0 * (double)p.P * (256 + (p.Xb ? 256 : 0) + (dw ? p.ky : 0) + (p.Wp ? p.ndx : 0)) + (dw ? 4.0 * grid * 256.0 * (p.ldk + 1) : 0.0);
  TimingScope ts_(dw ? "narrow_bwd" : "narrow_dx", 0, 301, p.P, dw ? 256 : p.ndx, dw ? p.ky : 256, (p.Wp ? 1 : 0) + (dw ? 1 : 0), s, bytes);
  if (!dw && p.Xb) hipLaunchKernelGGL((narrow_bwd_kernel<true, false, true>), dim3(grid), dim3(WS_THREADS), NB_LDS_DX, s, p, (int)tpw);
  else if (!dw) hipLaunchKernelGGL((narrow_bwd_kernel<true, false, false>), dim3(grid), dim3(WS_THREADS), NB_LDS_DX, s, p, (int)tpw);
  else if (p.Wp) hipLaunchKernelGGL((narrow_bwd_kernel<true, true, false>), dim3(grid), dim3(WS_THREADS), NB_LDS_DX, s, p, (int)tpw);
  else hipLaunchKernelGGL((narrow_bwd_kernel<false, true, false>), dim3(grid), dim3(WS_THREADS), NB_LDS_NODX, s, p, (int)tpw);
  CNR_LAUNCH_CHECK("narrow_bwd");
}

}  // namespace cnr
