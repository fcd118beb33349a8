// Weight-gradient GEMMs for gfx950: FP32-MFMA tail tiles, split-bf16 and split-f16 256 x 256 tiles, skinny strips, scale reduction.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"

namespace cnr {

// ================================================================================================
// weight-gradient GEMM:  dW[n][k] = sum_pt X[pt][n] * Y[pt][k]
// 8 waves as WR x WC, each wave (MT*32) x (KT*32); block tile TN x TK; points streamed 16 at a time.
// ================================================================================================
// points per slab: 16 for the square tile; the skinny tail tiles are latency-bound streams, so they take as many as LDS holds
constexpr int dw_bp(int tn, int tk) { return tn + tk <= 288 ? 64 : (tn + tk <= 320 ? 48 : 16); }

template <int WR, int WC, int MT, int KT, int KINDS = 0>
__global__ __launch_bounds__(512) void dw_gemm_kernel(const DwGemm g, int n0, int k0) {
  constexpr int TN = WR * MT * 32, TK = WC * KT * 32;
  constexpr int DW_BP = dw_bp(TN, TK);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [2][16][TN]
  float* Ys = smem + 2 * DW_BP * TN;      // [2][16][TK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const long chunk = blockIdx.x;
  const long p_begin = chunk * g.chunk_pts;
  long p_end = p_begin + g.chunk_pts;
  if (p_end > g.P) p_end = g.P;
  const int nslab_pair = p_end > p_begin ? (int)((p_end - p_begin + DW_BP - 1) / DW_BP) : 0;
  const int nslab = nslab_pair * g.npairs;
  constexpr int NX = (DW_BP * TN / 4 + 511) / 512, NY = (DW_BP * TK / 4 + 511) / 512;
  constexpr bool NX_EXACT = (DW_BP * TN / 4) % 512 == 0, NY_EXACT = (DW_BP * TK / 4) % 512 == 0;

  f32x16 acc[MT][KT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f4 rxa[NX], rxb[NX], rya[NY], ryb[NY];
  bool okx[NX], oky[NY];
  int staged_pair = 0;
  float csum = 0.0f;
  const bool want_colsum = g.colsum != nullptr && k0 == 0;

  // local views with the kinds pinned for the specialised instantiations (KINDS: 0 interpreted, 1 single pair all-direct, 2 SDF layer 0)
  View vX0 = g.X[0], vY0 = g.Y[0], vX1 = g.X[1], vY1 = g.Y[1];
  if (KINDS == 1) { vX0.kind = VK_DIRECT; vY0.kind = VK_DIRECT; vX0.scale = 1.0f; vY0.scale = 1.0f; }
  if (KINDS == 2) { vX0.kind = VK_DIRECT; vY0.kind = VK_DIRECT; vX1.kind = VK_SIGMUL; vY1.kind = VK_DIRECT;
                    vX0.scale = 1.0f; vY0.scale = 1.0f; vX1.scale = 1.0f; vY1.scale = 1.0f; }
#define DW_FETCH_(X_, Y_)                                                                                 \
  {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                      \
      const int idx = tid + i * 512;                                                                      \
      if (NX_EXACT || idx < DW_BP * TN / 4) {                                                             \
        const int pl = idx / (TN / 4), c4 = idx % (TN / 4);                                               \
        long pt = pbase_ + pl;                                                                            \
        okx[i] = pt < p_end;                                                                              \
        if (!okx[i]) pt = p_end - 1;                                                                      \
        const Raw4 q_ = view_fetch4(X_, pt, n0 + c4 * 4);                                                 \
        rxa[i] = q_.a; rxb[i] = q_.b;                                                                     \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NY; ++i) {                                                      \
      const int idx = tid + i * 512;                                                                      \
      if (NY_EXACT || idx < DW_BP * TK / 4) {                                                             \
        const int pl = idx / (TK / 4), c4 = idx % (TK / 4);                                               \
        long pt = pbase_ + pl;                                                                            \
        oky[i] = pt < p_end;                                                                              \
        if (!oky[i]) pt = p_end - 1;                                                                      \
        const Raw4 q_ = view_fetch4(Y_, pt, k0 + c4 * 4);                                                 \
        rya[i] = q_.a; ryb[i] = q_.b;                                                                     \
      }                                                                                                   \
    }                                                                                                     \
  }
#define DW_LOAD_SLAB(s_)                                                                                  \
  {                                                                                                       \
    const int pair_ = KINDS == 1 ? 0 : (s_) / nslab_pair;                                                 \
    staged_pair = pair_;                                                                                  \
    const long pbase_ = p_begin + (long)((s_) - pair_ * nslab_pair) * DW_BP;                              \
    if (pair_ == 0) DW_FETCH_(vX0, vY0) else DW_FETCH_(vX1, vY1)                                          \
  }
#define DW_FINISH_(X_, Y_, buf_)                                                                          \
  {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                      \
      const int idx = tid + i * 512;                                                                      \
      if (NX_EXACT || idx < DW_BP * TN / 4) {                                                             \
        Raw4 q_; q_.a = rxa[i]; q_.b = rxb[i];                                                            \
        const f4 v_ = view_finish4(X_, q_, n0 + (idx % (TN / 4)) * 4);                                    \
        *reinterpret_cast<f4*>(Xs + (buf_) * DW_BP * TN + idx * 4) = okx[i] ? v_ : z4_;                   \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NY; ++i) {                                                      \
      const int idx = tid + i * 512;                                                                      \
      if (NY_EXACT || idx < DW_BP * TK / 4) {                                                             \
        Raw4 q_; q_.a = rya[i]; q_.b = ryb[i];                                                            \
        const f4 v_ = view_finish4(Y_, q_, k0 + (idx % (TK / 4)) * 4);                                    \
        *reinterpret_cast<f4*>(Ys + (buf_) * DW_BP * TK + idx * 4) = oky[i] ? v_ : z4_;                   \
      }                                                                                                   \
    }                                                                                                     \
  }
#define DW_STORE_SLAB(buf_)                                                                               \
  {                                                                                                       \
    const f4 z4_ = {0.f, 0.f, 0.f, 0.f};                                                                  \
    if (staged_pair == 0) DW_FINISH_(vX0, vY0, buf_) else DW_FINISH_(vX1, vY1, buf_)                      \
  }

  if (nslab > 0) {
    DW_LOAD_SLAB(0)
    DW_STORE_SLAB(0)
  }
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) DW_LOAD_SLAB(s + 1)
    const float* Xb = Xs + buf * DW_BP * TN + (lane >> 5) * TN + wr * MT * 32 + (lane & 31);
    const float* Yb = Ys + buf * DW_BP * TK + (lane >> 5) * TK + wc * KT * 32 + (lane & 31);
#pragma unroll
    for (int st = 0; st < DW_BP / 2; ++st) {
      float a[MT], b[KT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = Xb[st * 2 * TN + i * 32];
#pragma unroll
      for (int j = 0; j < KT; ++j) b[j] = Yb[st * 2 * TK + j * 32];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (want_colsum && s < nslab_pair && tid < TN) {   // bias gradient: column sums of the first X operand
      const float* xc = Xs + buf * DW_BP * TN + tid;
#pragma unroll
      for (int pl = 0; pl < DW_BP; ++pl) csum += xc[pl * TN];
    }
    if (s + 1 < nslab) DW_STORE_SLAB(buf ^ 1)
    __syncthreads();
  }
#undef DW_LOAD_SLAB
#undef DW_STORE_SLAB
#undef DW_FETCH_
#undef DW_FINISH_

  float* out = g.partial + chunk * (long)g.Npad * g.ldk;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int kk = k0 + wc * KT * 32 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wr * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.Npad && kk < g.ldk) out[(long)n * g.ldk + kk] = acc[i][j][r];
      }
    }
  if (want_colsum && tid < TN && n0 + tid < g.Npad) g.colsum[chunk * g.Npad + n0 + tid] = csum;
}

template <int WR, int WC, int MT, int KT, int KINDS = 0>
static void launch_dw(const DwGemm& g, int n0, int k0, cnr_stream s) {
  constexpr int TN = WR * MT * 32, TK = WC * KT * 32;
  constexpr int DW_BP = dw_bp(TN, TK);
  const size_t lds = (size_t)(2 * DW_BP * (TN + TK)) * sizeof(float);
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_gemm_kernel<WR, WC, MT, KT, KINDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int tn_ = (g.N - n0) < TN ? (g.N - n0) : TN, tk_ = (g.K - k0) < TK ? (g.K - k0) : TK;
  TimingScope ts_("dw_gemm", 1, WR * 1000 + WC * 100 + MT * 10 + KT, g.P, tn_, tk_, g.npairs, s, dw_gemm_bytes(g, tn_, tk_));
  hipLaunchKernelGGL((dw_gemm_kernel<WR, WC, MT, KT, KINDS>), dim3(g.nchunk), dim3(512), lds, s, g, n0, k0);
}

// ================================================================================================
// weight-gradient GEMM, 256 x 256 output tile, split-bf16 matrix cores
//
// Same output-stationary structure as dw_gemm_kernel<4,2,2,4> (128 accumulator registers per wave, one workgroup per point
// chunk), but each fp32 operand element is split into three bf16 terms x = x1 + x2 + x3 (8 + 8 + 8 significand bits, the fp32
// exponent range is kept, so no scaling is needed along the contraction over points) and each product is evaluated as six
// v_mfma_f32_32x32x16_bf16 (x1y1 + x1y2 + x2y1 + x2y2 + x1y3 + x3y1, small terms first; the dropped terms are < 2^-25
// relative).  6 x 32 cycles per 32x32x16 block against 8 x 64 for v_mfma_f32_32x32x2_f32: 2.7x the matrix rate at fp32 accuracy.
//
// LDS: per operand and plane the 16-point slab is stored as [half h][j = n % 4][c = n / 4][8 points] bf16 with 1088-byte
// j-regions: the staging threads (4 columns x 4 points each) write 8-byte point quads, adjacent lanes adjacent quads
// (conflict-free), and an MFMA lane reads the 16 bytes of its row n / point half h (conflict-free: 16 lanes cover 16
// distinct 16-byte bank groups).
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int DX_JREG = 1088;                 // bytes per j-region (64 columns x 16 bytes + 64 bytes of bank rotation)
constexpr int DX_HALF = 4 * DX_JREG;          // one point half (8 points) of one plane
constexpr int DX_PLANE = 2 * DX_HALF;         // one bf16 plane of a 16-point x 256-column slab
constexpr int DX_OPER = 3 * DX_PLANE;         // three planes
constexpr int DX_BUF = 2 * DX_OPER;           // X and Y

__device__ __forceinline__ void dx_split3(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
  h1 = (__bf16)x;
  float r = x - (float)h1;
  h2 = (__bf16)r;
  r = r - (float)h2;
  h3 = (__bf16)r;
}

// XK0 / YK0 / XK1 / YK1 >= 0 pin the view kinds of the operand pairs at compile time (-1 = interpreted at run time)
template <int XK0, int YK0, int XK1, int YK1>
__global__ __launch_bounds__(512, 1) void dw_gemm_bx_kernel(const DwGemm g_in, int n0, int k0) {
  DwGemm g = g_in;
  if (XK0 >= 0) g.X[0].kind = XK0;
  if (YK0 >= 0) g.Y[0].kind = YK0;
  if (XK1 >= 0) g.X[1].kind = XK1;
  if (YK1 >= 0) g.Y[1].kind = YK1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;               // 4 x 2 waves, 64 x 128 outputs each
  // 16-point slabs are dealt round-robin to the workgroups (slab = i * nchunk + chunk): at any moment the CUs read one
  // contiguous stretch of X and Y, which spreads over all HBM channels; the partial sums stay in a fixed order
  const long chunk = blockIdx.x;
  const long p_end = g.P;
  const long total_slabs = (g.P + 15) / 16;
  const int nslab_pair = chunk < total_slabs ? (int)((total_slabs - chunk + g.nchunk - 1) / g.nchunk) : 0;
  const int nslab = nslab_pair * g.npairs;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // staging role: threads 0..255 stage X, 256..511 stage Y; each 4 columns x 4 points
  const bool is_y = tid >= 256;
  const int st = tid & 255;
  const int qlo = st & 1, c4 = (st >> 1) & 63, qhi = st >> 7;
  const int q = qhi * 2 + qlo;                          // point quad of the slab
  const int scol = (is_y ? k0 : n0) + c4 * 4;
  unsigned char* const sdst = smem_d + (is_y ? DX_OPER : 0) + qhi * DX_HALF + c4 * 16 + qlo * 8;
  f4 ra[4], rb[4];
  bool okp[4];
  int staged_pair = 0;
  f4 csum = {0.f, 0.f, 0.f, 0.f};
  const bool want_colsum = g.colsum != nullptr && k0 == 0;

#define DX_FETCH_(V_)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
    long pt = pbase_ + i;                                                                  \
    okp[i] = pt < p_end;                                                                   \
    if (!okp[i]) pt = p_end - 1;                                                           \
    const Raw4 q_ = view_fetch4(V_, pt, scol);                                             \
    ra[i] = q_.a; rb[i] = q_.b;                                                            \
  }
#define DX_LOAD_SLAB(s_)                                                                   \
  {                                                                                        \
    const int pair_ = (s_) / nslab_pair;                                                   \
    staged_pair = pair_;                                                                   \
    const long pbase_ = ((long)((s_) - pair_ * nslab_pair) * g.nchunk + chunk) * 16 + q * 4; \
    if (pair_ == 0) { if (is_y) { DX_FETCH_(g.Y[0]) } else { DX_FETCH_(g.X[0]) } }         \
    else { if (is_y) { DX_FETCH_(g.Y[1]) } else { DX_FETCH_(g.X[1]) } }                    \
  }
#define DX_FINISH_(V_)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
    Raw4 q_; q_.a = ra[i]; q_.b = rb[i];                                                   \
    v_[i] = okp[i] ? view_finish4(V_, q_, scol) : z4_;                                     \
  }
#define DX_STORE_SLAB(buf_)                                                                \
  {                                                                                        \
    const f4 z4_ = {0.f, 0.f, 0.f, 0.f};                                                   \
    f4 v_[4];                                                                              \
    if (staged_pair == 0) { if (is_y) { DX_FINISH_(g.Y[0]) } else { DX_FINISH_(g.X[0]) } } \
    else { if (is_y) { DX_FINISH_(g.Y[1]) } else { DX_FINISH_(g.X[1]) } }                  \
    if (want_colsum && !is_y && staged_pair == 0) {                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { csum.x += v_[i].x; csum.y += v_[i].y; csum.z += v_[i].z; csum.w += v_[i].w; } \
    }                                                                                      \
    unsigned char* d_ = sdst + (buf_) * DX_BUF;                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
      bf16x4 h1, h2, h3;                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { __bf16 a_, b_, c_; dx_split3(v_[i][j], a_, b_, c_); h1[i] = a_; h2[i] = b_; h3[i] = c_; } \
      *reinterpret_cast<bf16x4*>(d_ + j * DX_JREG) = h1;                                   \
      *reinterpret_cast<bf16x4*>(d_ + DX_PLANE + j * DX_JREG) = h2;                        \
      *reinterpret_cast<bf16x4*>(d_ + 2 * DX_PLANE + j * DX_JREG) = h3;                    \
    }                                                                                      \
  }

  // The X-staging waves (0..3) and the Y-staging waves (4..7) share the SIMDs pairwise and run half an iteration out of
  // phase: while one wave of a SIMD issues its MFMAs, the other converts and stores its part of the next slab, so the
  // matrix pipe and the VALU / LDS-store path overlap.  X waves: load(s+1) | MFMA(s) | store(s+1);  Y waves: store(s+1),
  // load(s+2) | MFMA(s).  One barrier per slab.
  if (nslab > 0) {
    DX_LOAD_SLAB(0)
    DX_STORE_SLAB(0)
    if (is_y && nslab > 1) DX_LOAD_SLAB(1)
  }
  __syncthreads();
  // operand fragment addresses of this lane: row / column (lane & 31) of each 32-wide tile, point half (lane >> 5)
  const int ln = lane & 31, lh = lane >> 5;
  const int xoff = lh * DX_HALF + (ln & 3) * DX_JREG + (wr * 16 + (ln >> 2)) * 16;              // + i * 8 * 16 per n-tile
  const int yoff = DX_OPER + lh * DX_HALF + (ln & 3) * DX_JREG + (wc * 32 + (ln >> 2)) * 16;    // + j * 8 * 16 per k-tile
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (!is_y) {
      if (s + 1 < nslab) DX_LOAD_SLAB(s + 1)
    } else if (s + 1 < nslab) {
      DX_STORE_SLAB(buf ^ 1)
      if (s + 2 < nslab) DX_LOAD_SLAB(s + 2)
    }
    const unsigned char* B_ = smem_d + buf * DX_BUF;
    bf16x8 a[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(B_ + xoff + p * DX_PLANE + i * 128);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(B_ + yoff + p * DX_PLANE + j * 128);
      // the two row tiles alternate so that back-to-back MFMAs never depend on each other
      f32x16 c0 = acc[0][j], c1 = acc[1][j];
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][2], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][2], b[0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[2], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b[2], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], b[1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], b[1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], b[0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b[1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b[0], c1, 0, 0, 0);
      acc[0][j] = c0; acc[1][j] = c1;
    }
    if (!is_y && s + 1 < nslab) DX_STORE_SLAB(buf ^ 1)
    __syncthreads();
  }
#undef DX_LOAD_SLAB
#undef DX_STORE_SLAB
#undef DX_FETCH_
#undef DX_FINISH_

  float* out = g.partial + chunk * (long)g.Npad * g.ldk;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + wc * 128 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.Npad && kk < g.ldk) out[(long)n * g.ldk + kk] = acc[i][j][r];
      }
    }
  if (want_colsum) {   // bias gradient: the four point-quad owners of a column group add their sums in a fixed order
    float* cs = reinterpret_cast<float*>(smem_d);          // [4 quads][256 columns]; the slab buffers are dead now
    if (!is_y) *reinterpret_cast<f4*>(cs + q * 256 + c4 * 4) = csum;
    __syncthreads();
    if (tid < 256 && n0 + tid < g.Npad) g.colsum[chunk * g.Npad + n0 + tid] = ((cs[tid] + cs[256 + tid]) + cs[512 + tid]) + cs[768 + tid];
  }
}

template <int XK0, int YK0, int XK1, int YK1>
static void launch_dw_bx_t(const DwGemm& g, int n0, int k0, cnr_stream s) {
  const size_t lds = (size_t)2 * DX_BUF;
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_gemm_bx_kernel<XK0, YK0, XK1, YK1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int tn_ = (g.N - n0) < 256 ? (g.N - n0) : 256, tk_ = (g.K - k0) < 256 ? (g.K - k0) : 256;
  TimingScope ts_("dw_gemm_bx", 1, 4224, g.P, tn_, tk_, g.npairs, s, dw_gemm_bytes(g, tn_, tk_));
  hipLaunchKernelGGL((dw_gemm_bx_kernel<XK0, YK0, XK1, YK1>), dim3(g.nchunk), dim3(512), lds, s, g, n0, k0);
}

static void launch_dw_bx(const DwGemm& g, int n0, int k0, cnr_stream s) {
  const int x0 = g.X[0].kind, y0 = g.Y[0].kind, x1 = g.npairs > 1 ? g.X[1].kind : -1, y1 = g.npairs > 1 ? g.Y[1].kind : -1;
#define DX_CASE(A_, B_, C_, D_) if (x0 == A_ && y0 == B_ && x1 == C_ && y1 == D_) { launch_dw_bx_t<A_, B_, C_, D_>(g, n0, k0, s); return; }
  // the operand combinations of the render plan (cnr_plan.cpp): MLP layers, SDF value + gradient-chain pairs
  DX_CASE(VK_DIRECT, VK_DIRECT, -1, -1)
  DX_CASE(VK_DIRECT, VK_SOFTPLUS, -1, -1)
  DX_CASE(VK_DIRECT, VK_SOFTPLUS, VK_SIGMUL, VK_DIRECT)
  DX_CASE(VK_DIRECT, VK_SOFTPLUS, VK_SIGMUL_ROW, VK_DIRECT)
  DX_CASE(VK_DIRECT, VK_SOFTPLUS, VK_CONST_COL0, VK_DIRECT)
#undef DX_CASE
  launch_dw_bx_t<-1, -1, -1, -1>(g, n0, k0, s);
}

// ================================================================================================
// weight-gradient GEMM, 256 x 256 output tile, split-f16 matrix cores (three MFMAs per product instead of six)
//
// Along the contraction (points) a scale can only be used if it cancels per point: X'[pt] = X[pt] * sx[pt] (the power of
// two that lifts the row into the top f16 binade -- emitted for free by the layer GEMM that consumed the same operand,
// LayerGemm::rs_out) and Y'[pt] = Y[pt] * 2^G / sx[pt], so X'^T Y' = 2^G X^T Y exactly.  G = 1 + min over the workgroup's points of
// log2(sx * sy) keeps every Y' row below 2^15; points whose product is far below the largest one lose relative
// precision in Y' but their absolute error stays below 2^-40 of the largest term.  Both operands are split hi + lo
// (11 + 11 bits) and x1 y2 + x2 y1 + x1 y1 is accumulated in fp32; the result is scaled back by 2^-G (exact).
// Structure, LDS layout and wave phase shift as dw_gemm_bx_kernel (two planes per operand instead of three).
// ================================================================================================
constexpr int DH_OPER = 2 * DX_PLANE;
constexpr int DH_BUF = 2 * DH_OPER;

// 2^G / sx for a power-of-two sx > 0 by exponent arithmetic (0 stays 0, NaN stays NaN, underflow flushes to 0)
__device__ __forceinline__ float dh_yscale(float sx, int G) {
  const unsigned bits = __float_as_uint(sx);
  const int field = G - (int)((bits >> 23) & 0xff) + 254;      // biased exponent of 2^(G - log2 sx)
  const float r = __uint_as_float((unsigned)(field < 1 ? 0 : (field > 254 ? 254 : field)) << 23);
  return sx > 0.0f ? (field < 1 ? 0.0f : r) : sx;
}

template <int XK0, int YK0, int XK1, int YK1>
__global__ __launch_bounds__(512, 1) void dw_gemm_hx_kernel(const DwGemm g_in, int n0, int k0) {
  DwGemm g = g_in;
  if (XK0 >= 0) g.X[0].kind = XK0;
  if (YK0 >= 0) g.Y[0].kind = YK0;
  if (XK1 >= 0) g.X[1].kind = XK1;
  if (YK1 >= 0) g.Y[1].kind = YK1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const long chunk = blockIdx.x;
  const long p_end = g.P;
  const long total_slabs = (g.P + 15) / 16;
  const int nslab_pair = chunk < total_slabs ? (int)((total_slabs - chunk + g.nchunk - 1) / g.nchunk) : 0;
  const int nslab = nslab_pair * g.npairs;
  // exponent of this workgroup's slice of the point range: G = 1 + min log2(sx * sy) over its own points (both pairs, all-zero rows
  // excluded): no separate reduction launch, and a slice of small-magnitude points keeps its own dynamic range
  int G = 0x7f7f7f7f;
  {
    const int npts = nslab_pair * 16;
    auto scan = [&](const float* sxp, const float* syp) {   // (no run-time index into g: the struct must stay in registers)
      for (int i = tid; i < npts; i += 512) {
        const long pt = ((long)(i >> 4) * g.nchunk + chunk) * 16 + (i & 15);
        if (pt < p_end) {
          const float a = sxp[pt], b = syp[pt];   // powers of two (or 0 / NaN): exponent = biased exponent field - 127
          const int e = (int)((__float_as_uint(a) >> 23) & 0xff) + (int)((__float_as_uint(b) >> 23) & 0xff) - 254 + 1;
          if (a > 0.0f && b > 0.0f && e < G) G = e;
        }
      }
    };
    scan(g.sx[0], g.sy[0]);
    if (g.npairs > 1) scan(g.sx[1], g.sy[1]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(G, d); G = o < G ? o : G; }
    int* red = reinterpret_cast<int*>(smem_d);
    if (lane == 0) red[wave] = G;
    cnr_lds_barrier();
#pragma unroll
    for (int w = 0; w < 8; ++w) { const int o = red[w]; G = o < G ? o : G; }
    cnr_lds_barrier();
    G = __builtin_amdgcn_readfirstlane(G);                // (uniform: keep it in a scalar register)
    if (G > 250 || G < -250) G = 0;                       // no point with two non-zero rows: everything is zero anyway
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const bool is_y = tid >= 256;
  const int st = tid & 255;
  const int qlo = st & 1, c4 = (st >> 1) & 63, qhi = st >> 7;
  const int q = qhi * 2 + qlo;
  const int scol = (is_y ? k0 : n0) + c4 * 4;
  unsigned char* const sdst = smem_d + (is_y ? DH_OPER : 0) + qhi * DX_HALF + c4 * 16 + qlo * 8;
  f4 ra[4], rb[4];
  float rsx[4];                                           // sx of the 4 points this thread stages
  bool okp[4];
  int staged_pair = 0;
  f4 csum = {0.f, 0.f, 0.f, 0.f};
  const bool want_colsum = g.colsum != nullptr && k0 == 0;

#define DH_FETCH_(V_)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
    long pt = pbase_ + i;                                                                  \
    okp[i] = pt < p_end;                                                                   \
    if (!okp[i]) pt = p_end - 1;                                                           \
    const Raw4 q_ = view_fetch4(V_, pt, scol);                                             \
    ra[i] = q_.a; rb[i] = q_.b;                                                            \
    rsx[i] = sxp_[pt];                                                                     \
  }
  constexpr bool kOnePair = XK0 >= 0 && XK1 < 0;   // specialised single-pair instantiation: the pair index folds to 0
#define DH_LOAD_SLAB(s_)                                                                   \
  {                                                                                        \
    const int pair_ = kOnePair ? 0 : (s_) / nslab_pair;                                    \
    staged_pair = pair_;                                                                   \
    const long pbase_ = ((long)((s_) - pair_ * nslab_pair) * g.nchunk + chunk) * 16 + q * 4; \
    const float* sxp_ = pair_ == 0 ? g.sx[0] : g.sx[1];                                    \
    if (pair_ == 0) { if (is_y) { DH_FETCH_(g.Y[0]) } else { DH_FETCH_(g.X[0]) } }         \
    else { if (is_y) { DH_FETCH_(g.Y[1]) } else { DH_FETCH_(g.X[1]) } }                    \
  }
#define DH_FINISH_(V_)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
    Raw4 q_; q_.a = ra[i]; q_.b = rb[i];                                                   \
    v_[i] = okp[i] ? view_finish4(V_, q_, scol) : z4_;                                     \
  }
#define DH_STORE_SLAB(buf_)                                                                \
  {                                                                                        \
    const f4 z4_ = {0.f, 0.f, 0.f, 0.f};                                                   \
    f4 v_[4];                                                                              \
    if (staged_pair == 0) { if (is_y) { DH_FINISH_(g.Y[0]) } else { DH_FINISH_(g.X[0]) } } \
    else { if (is_y) { DH_FINISH_(g.Y[1]) } else { DH_FINISH_(g.X[1]) } }                  \
    if (want_colsum && !is_y && staged_pair == 0) {                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { csum.x += v_[i].x; csum.y += v_[i].y; csum.z += v_[i].z; csum.w += v_[i].w; } \
    }                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
      float f_ = rsx[i];                                                                   \
      if (is_y) f_ = dh_yscale(f_, G);                                                      \
      v_[i].x *= f_; v_[i].y *= f_; v_[i].z *= f_; v_[i].w *= f_;                          \
    }                                                                                      \
    unsigned char* d_ = sdst + (buf_) * DH_BUF;                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
      f16x4 h1, h2;                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { const float x_ = v_[i][j]; h1[i] = (_Float16)x_; h2[i] = (_Float16)(x_ - (float)h1[i]); } \
      *reinterpret_cast<f16x4*>(d_ + j * DX_JREG) = h1;                                    \
      *reinterpret_cast<f16x4*>(d_ + DX_PLANE + j * DX_JREG) = h2;                         \
    }                                                                                      \
  }

  if (nslab > 0) {
    DH_LOAD_SLAB(0)
    DH_STORE_SLAB(0)
    if (is_y && nslab > 1) DH_LOAD_SLAB(1)
  }
  cnr_lds_barrier();
  const int ln = lane & 31, lh = lane >> 5;
  const int xoff = lh * DX_HALF + (ln & 3) * DX_JREG + (wr * 16 + (ln >> 2)) * 16;
  const int yoff = DH_OPER + lh * DX_HALF + (ln & 3) * DX_JREG + (wc * 32 + (ln >> 2)) * 16;
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (!is_y) {
      if (s + 1 < nslab) DH_LOAD_SLAB(s + 1)
    } else if (s + 1 < nslab) {
      DH_STORE_SLAB(buf ^ 1)
      if (s + 2 < nslab) DH_LOAD_SLAB(s + 2)
    }
    const unsigned char* B_ = smem_d + buf * DH_BUF;
    f16x8 a[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[i][p] = *reinterpret_cast<const f16x8*>(B_ + xoff + p * DX_PLANE + i * 128);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x8 b[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) b[p] = *reinterpret_cast<const f16x8*>(B_ + yoff + p * DX_PLANE + j * 128);
      f32x16 c0 = acc[0][j], c1 = acc[1][j];
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], b[1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], b[1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1], b[0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], b[0], c1, 0, 0, 0);
      acc[0][j] = c0; acc[1][j] = c1;
    }
    if (!is_y && s + 1 < nslab) DH_STORE_SLAB(buf ^ 1)
    cnr_lds_barrier();
  }
#undef DH_LOAD_SLAB
#undef DH_STORE_SLAB
#undef DH_FETCH_
#undef DH_FINISH_

  // undo 2^G in two exact steps (G can exceed the fp32 exponent range of a single factor)
  const float u1 = ldexpf(1.0f, -(G / 2)), u2 = ldexpf(1.0f, -(G - G / 2));
  float* out = g.partial + chunk * (long)g.Npad * g.ldk;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + wc * 128 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.Npad && kk < g.ldk) out[(long)n * g.ldk + kk] = acc[i][j][r] * u1 * u2;
      }
    }
  if (want_colsum) {
    float* cs = reinterpret_cast<float*>(smem_d);
    if (!is_y) *reinterpret_cast<f4*>(cs + q * 256 + c4 * 4) = csum;
    cnr_lds_barrier();
    if (tid < 256 && n0 + tid < g.Npad) g.colsum[chunk * g.Npad + n0 + tid] = ((cs[tid] + cs[256 + tid]) + cs[512 + tid]) + cs[768 + tid];
  }
}

template <int XK0, int YK0, int XK1, int YK1>
static void launch_dw_hx_t(const DwGemm& g, int n0, int k0, cnr_stream s) {
  const size_t lds = (size_t)2 * DH_BUF;
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_gemm_hx_kernel<XK0, YK0, XK1, YK1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int tn_ = (g.N - n0) < 256 ? (g.N - n0) : 256, tk_ = (g.K - k0) < 256 ? (g.K - k0) : 256;
  TimingScope ts_("dw_gemm_hx", 1, 4224, g.P, tn_, tk_, g.npairs, s, dw_gemm_bytes(g, tn_, tk_));
  hipLaunchKernelGGL((dw_gemm_hx_kernel<XK0, YK0, XK1, YK1>), dim3(g.nchunk), dim3(512), lds, s, g, n0, k0);
}

static void launch_dw_hx(const DwGemm& g, int n0, int k0, cnr_stream s) {
  const int x0 = g.X[0].kind, y0 = g.Y[0].kind, x1 = g.npairs > 1 ? g.X[1].kind : -1, y1 = g.npairs > 1 ? g.Y[1].kind : -1;
#define DH_CASE(A_, B_, C_, D_) if (x0 == A_ && y0 == B_ && x1 == C_ && y1 == D_) { launch_dw_hx_t<A_, B_, C_, D_>(g, n0, k0, s); return; }
  DH_CASE(VK_DIRECT, VK_DIRECT, -1, -1)
  DH_CASE(VK_DIRECT, VK_SOFTPLUS, -1, -1)
  DH_CASE(VK_DIRECT, VK_SOFTPLUS, VK_SIGMUL, VK_DIRECT)
  DH_CASE(VK_DIRECT, VK_SOFTPLUS, VK_SIGMUL_ROW, VK_DIRECT)
#undef DH_CASE
  launch_dw_hx_t<-1, -1, -1, -1>(g, n0, k0, s);
}

// ================================================================================================
// weight-gradient strips with one very narrow side (<= 8 columns): the 3 / 6 extra input columns of the colour and
// relight nets, the rgb / sdf output rows.  Pure HBM streams (<= 12 FLOP per loaded byte): no matrix cores, no LDS
// staging.  One workgroup per point chunk, 16 waves, each wave walks its own points (slabs dealt round-robin as in
// dw_gemm_bx); a lane owns 4 columns of the wide operand x all narrow columns in registers; the 8 waves are folded
// through LDS in a fixed order.  NARROW_X: the narrow operand is X (rows of dW), otherwise Y (columns of dW).
// ================================================================================================
constexpr int SK_WAVES = 8;

// KINDS: 0 = interpreted views; 1 = every view VK_DIRECT with scale 1 (4 of the 5 strips of a step); 2 = the sdf row of the top
// SDF layer (zbar x softplus(z) + unit vector x qbar).  With the kinds pinned the interpreted prologue folds away.
template <bool NARROW_X, bool NG2, int KINDS>
__global__ __launch_bounds__(SK_WAVES * 64) void dw_skinny_kernel(const DwGemm g_in, int n0, int k0, int ncnt) {
  DwGemm g = g_in;
  if (KINDS == 1) {
    g.X[0].kind = VK_DIRECT; g.Y[0].kind = VK_DIRECT; g.X[1].kind = VK_DIRECT; g.Y[1].kind = VK_DIRECT;
    g.X[0].scale = 1.0f; g.Y[0].scale = 1.0f; g.X[1].scale = 1.0f; g.Y[1].scale = 1.0f;
  } else if (KINDS == 2) {
    g.X[0].kind = VK_DIRECT; g.Y[0].kind = VK_SOFTPLUS; g.X[1].kind = VK_CONST_COL0; g.Y[1].kind = VK_DIRECT;
    g.X[0].scale = 1.0f; g.Y[1].scale = 1.0f;
  }
  constexpr int SK_UNROLL = NG2 ? 4 : 8;
  extern __shared__ __attribute__((aligned(16))) float smem_k[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long chunk = blockIdx.x;
  const int wide0 = NARROW_X ? k0 : n0, narrow0 = NARROW_X ? n0 : k0;
  const int wide_lim = NARROW_X ? g.ldk : g.Npad;       // wide columns beyond the padded extent are neither read nor written
  const int wcol = wide0 + lane * 4;
  const bool wlive = wcol < wide_lim;
  f4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j].x = 0.f; acc[j].y = 0.f; acc[j].z = 0.f; acc[j].w = 0.f; }
  f4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};       // column sums of the narrow X operand (bias gradient)
  const bool want_colsum = NARROW_X && g.colsum != nullptr && k0 == 0;
  // points of this workgroup: 16-point slabs chunk, chunk + nchunk, ...; wave w takes points w and w + 8 of each slab
  const long total_slabs = (g.P + 15) / 16;
  const long my_slabs = chunk < total_slabs ? (total_slabs - chunk + g.nchunk - 1) / g.nchunk : 0;
  const long my_pts = my_slabs * (16 / SK_WAVES);       // per wave
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int pair = 0; pair < g.npairs; ++pair) {
    const View& Vn = NARROW_X ? (pair == 0 ? g.X[0] : g.X[1]) : (pair == 0 ? g.Y[0] : g.Y[1]);
    const View& Vw = NARROW_X ? (pair == 0 ? g.Y[0] : g.Y[1]) : (pair == 0 ? g.X[0] : g.X[1]);
    for (long i0 = 0; i0 < my_pts; i0 += SK_UNROLL) {
      // all loads of the round are issued before any of the (interpreted) view math: one memory round trip per round
      Raw4 wr[SK_UNROLL], nr0[SK_UNROLL], nr1[SK_UNROLL];
      unsigned okmask = 0;
#pragma unroll
      for (int u = 0; u < SK_UNROLL; ++u) {
        const long i = i0 + u;
        const long pt = ((i >> 1) * g.nchunk + chunk) * 16 + wave + SK_WAVES * (i & 1);
        const bool ok = i < my_pts && pt < g.P;
        okmask |= ok ? (1u << u) : 0u;
        const long ptc = ok ? pt : g.P - 1;
        wr[u] = view_fetch4(Vw, ptc, wlive ? wcol : wide0);
        nr0[u] = view_fetch4(Vn, ptc, narrow0);
        if (NG2) nr1[u] = view_fetch4(Vn, ptc, narrow0 + 4);
      }
#pragma unroll
      for (int u = 0; u < SK_UNROLL; ++u) {
        const bool ok = (okmask >> u) & 1;
        const f4 w = (ok && wlive) ? view_finish4(Vw, wr[u], wcol) : z4;
        const f4 a0 = ok ? view_finish4(Vn, nr0[u], narrow0) : z4;
        f4 a1 = z4;
        if (NG2 && ok) a1 = view_finish4(Vn, nr1[u], narrow0 + 4);
#pragma unroll
        for (int j = 0; j < (NG2 ? 8 : 4); ++j) {
          const float nj = j < 4 ? a0[j & 3] : a1[j & 3];
          acc[j].x += nj * w.x; acc[j].y += nj * w.y; acc[j].z += nj * w.z; acc[j].w += nj * w.w;
        }
        if (want_colsum && pair == 0) {
          cs0.x += a0.x; cs0.y += a0.y; cs0.z += a0.z; cs0.w += a0.w;
          cs1.x += a1.x; cs1.y += a1.y; cs1.z += a1.z; cs1.w += a1.w;
        }
      }
    }
  }
  // fold the waves in a fixed order: LDS [wave][8 narrow][256 wide]
  float* red = smem_k + (size_t)wave * 8 * 256 + lane * 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<f4*>(red + j * 256) = acc[j];
  float* csr = smem_k + (size_t)SK_WAVES * 8 * 256;
  if (want_colsum && lane == 0) {
    *reinterpret_cast<f4*>(csr + wave * 8) = cs0;
    *reinterpret_cast<f4*>(csr + wave * 8 + 4) = cs1;
  }
  __syncthreads();
  float* out = g.partial + chunk * (long)g.Npad * g.ldk;
  for (int e = tid; e < 8 * 256; e += SK_WAVES * 64) {
    const int j = e >> 8, c = e & 255;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) sum += smem_k[(size_t)w * 8 * 256 + e];
    const int n = NARROW_X ? narrow0 + j : wide0 + c, k = NARROW_X ? wide0 + c : narrow0 + j;
    if (j < ncnt && n < g.Npad && k < g.ldk) out[(long)n * g.ldk + k] = sum;
  }
  if (want_colsum && tid < ncnt) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) sum += csr[w * 8 + tid];
    if (n0 + tid < g.Npad) g.colsum[chunk * g.Npad + n0 + tid] = sum;
  }
}

template <bool NARROW_X, bool NG2, int KINDS>
static void launch_dw_skinny_td(const DwGemm& g, int n0, int k0, int ncnt, cnr_stream s) {
  const size_t lds = ((size_t)SK_WAVES * 8 * 256 + SK_WAVES * 8) * sizeof(float);
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_skinny_kernel<NARROW_X, NG2, KINDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int wide = NARROW_X ? ((g.K - k0) < 256 ? (g.K - k0) : 256) : ((g.N - n0) < 256 ? (g.N - n0) : 256);
  const int tn_ = NARROW_X ? ncnt : wide, tk_ = NARROW_X ? wide : ncnt;
  TimingScope ts_("dw_skinny", 1, NARROW_X ? 1 : 2, g.P, tn_, tk_, g.npairs, s, dw_gemm_bytes(g, tn_, tk_));
  hipLaunchKernelGGL((dw_skinny_kernel<NARROW_X, NG2, KINDS>), dim3(g.nchunk), dim3(SK_WAVES * 64), lds, s, g, n0, k0, ncnt);
}

template <bool NARROW_X, bool NG2>
static void launch_dw_skinny_t(const DwGemm& g, int n0, int k0, int ncnt, cnr_stream s) {
  bool direct = true;
  for (int i = 0; i < g.npairs; ++i) direct = direct && g.X[i].kind == VK_DIRECT && g.Y[i].kind == VK_DIRECT && g.X[i].scale == 1.0f && g.Y[i].scale == 1.0f;
  const bool sdf_row = NARROW_X && g.npairs == 2 && g.X[0].kind == VK_DIRECT && g.X[0].scale == 1.0f && g.Y[0].kind == VK_SOFTPLUS &&
                       g.X[1].kind == VK_CONST_COL0 && g.Y[1].kind == VK_DIRECT && g.Y[1].scale == 1.0f;
  if (direct) launch_dw_skinny_td<NARROW_X, NG2, 1>(g, n0, k0, ncnt, s);
  else if (sdf_row) launch_dw_skinny_td<NARROW_X, NG2, 2>(g, n0, k0, ncnt, s);
  else launch_dw_skinny_td<NARROW_X, NG2, 0>(g, n0, k0, ncnt, s);
}

template <bool NARROW_X>
static void launch_dw_skinny(const DwGemm& g, int n0, int k0, int ncnt, cnr_stream s) {
  if (ncnt > 4) launch_dw_skinny_t<NARROW_X, true>(g, n0, k0, ncnt, s);
  else launch_dw_skinny_t<NARROW_X, false>(g, n0, k0, ncnt, s);
}

void be_dw_gemm(const DwGemm& g, cnr_stream s) {
  // tile the [Npad x ldk] output: 256x256 main tiles, 256x64 column tails, 32x256 row tails
  const bool dw_fp32 = debug_flags().dw_fp32;   // debugging aid: FP32-MFMA kernel for the main tiles too
  const bool dw_bf16 = debug_flags().dw_bf16;   // debugging aid: split-bf16 kernel even when row scales are available
  for (int n0 = 0; n0 < g.N; n0 += 256) {
    const int nrem = g.N - n0;
    for (int k0 = 0; k0 < g.K;) {
      const int krem = g.K - k0;
      if (nrem > 32) {
        if (krem > 64 && g.skip_main) { k0 += 256; continue; }   // main tile formed by the fused layer + weight-gradient launch
        if (krem > 64) {
          // the gradient-chain pair of the top SDF layer is a unit vector at the sdf row: it adds nothing to other row tiles
          DwGemm gm = g;
          if (gm.npairs == 2 && gm.X[1].kind == VK_CONST_COL0) {
            const int hot = gm.X[1].math_split == (1 << 30) ? 0 : gm.X[1].math_split;
            if (hot < n0 || hot >= n0 + 256) gm.npairs = 1;
          }
          bool scaled = !dw_bf16 && gm.split_f16;
          for (int i = 0; i < gm.npairs; ++i) scaled = scaled && gm.sx[i] != nullptr && gm.sy[i] != nullptr;
          if (dw_fp32) launch_dw<4, 2, 2, 4>(g, n0, k0, s);
          else if (scaled) launch_dw_hx(gm, n0, k0, s);
          else launch_dw_bx(gm, n0, k0, s);
          k0 += 256;
        }
        else if (krem <= 8 && !dw_fp32 && !(g.colsum != nullptr && k0 == 0) && (k0 & 3) == 0) { launch_dw_skinny<false>(g, n0, k0, krem, s); k0 += 64; }
        else {   // 9..64-column strips (embedding inputs): FP32-MFMA tile, view kinds pinned for the two shapes of the plan
          const bool d0 = g.X[0].kind == VK_DIRECT && g.Y[0].kind == VK_DIRECT && g.X[0].scale == 1.0f && g.Y[0].scale == 1.0f;
          if (g.npairs == 1 && d0) launch_dw<8, 1, 1, 2, 1>(g, n0, k0, s);
          else if (g.npairs == 2 && d0 && g.X[1].kind == VK_SIGMUL && g.Y[1].kind == VK_DIRECT && g.X[1].scale == 1.0f && g.Y[1].scale == 1.0f)
            launch_dw<8, 1, 1, 2, 2>(g, n0, k0, s);
          else launch_dw<8, 1, 1, 2>(g, n0, k0, s);
          k0 += 64;
        }
      } else if (nrem <= 8 && !dw_fp32 && (n0 & 3) == 0) {
        launch_dw_skinny<true>(g, n0, k0, nrem, s); k0 += 256;
      } else {
        launch_dw<1, 8, 1, 1>(g, n0, k0, s); k0 += 256;
      }
    }
  }
  CNR_LAUNCH_CHECK("dw_gemm");
}


}  // namespace cnr
