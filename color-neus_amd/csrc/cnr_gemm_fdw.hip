// Layer GEMM + weight gradient in one launch for gfx950 (MI355X / CDNA4).
//
// Why: in the backward pass every 256 -> 256 layer used to cost two passes over the same rows: the layer launch (cotangent through W^T,
// or the second-order sweep through W) and, later, the weight-gradient GEMM that reads the launch's input AND the tensor its epilogue had
// already fetched as a side input (2 - 3 KB per point and layer again, 23 % of a training step).  Here the weight gradient is formed
// while both operands are on chip.
//
// The obstacle is the register file: a 256 x 256 layer as two f16 planes is 256 KB, a 256 x 256 fp32 accumulator another 256 KB, and a CU
// has 512 KB.  So a launch is split into COLUMN HALVES: a workgroup owns 128 output columns (weights: 128 KB) and the matching
// [256 x 128] block of dW (128 KB); the two halves of a point range run on the same XCD (blocks b and b + 8), so the second read of the
// input tile is an L2 hit.  Inside a workgroup the waves are specialised (one of each kind per SIMD):
//   waves 0..3 "P"  weight-stationary layer product of their 32 columns (3 x v_mfma_f32_32x32x16_f16 per k16 block, as cnr_gemm_ws.h),
//                   fused epilogue with 16-byte global accesses, and -- from the side inputs the epilogue fetched anyway -- the tile of the
//                   epilogue-side operand Ep, scaled and split hi / lo, written row-major ([point][column], one 8-byte store per plane and 4 columns) into LDS;
//   waves 4..7 "D"  stage the input tile S (HBM -> registers -> exact power-of-two row scale -> f16 hi / lo planes in LDS, the operand of
//                   the P waves), and accumulate C[s][e] += sum_pt S'[pt][s] * Ep'[pt][e] for the tile the P waves finished one
//                   iteration earlier: both fragments come out of the row-major planes by the LDS transpose read (ds_read_b64_tr_b16, two per
//                   fragment: fd_tr8 below); 64 x 128 outputs per wave = 128 accumulator registers.
// Per 32-point tile and SIMD: 48 + 48 MFMAs.  One workgroup barrier per tile; three input-tile buffers (stage t + 1 | product t | dW t - 1).
//
// Scaling of the split-f16 weight gradient (same scheme as dw_gemm_hx_kernel): S' = S * ss[pt] are the planes the layer product uses
// anyway; Ep' = Ep * 2^G / ss[pt], so S'^T Ep' = 2^G S^T Ep exactly.  G = 1 + min over the points seen so far of log2(ss * se) (se: the
// row scale of Ep saved by the forward pass) is a RUNNING minimum: when a tile lowers it the accumulators are rescaled by the exact
// power of two (wave-uniform, rare after the first tiles).  Fixed summation order per (range, half): bitwise deterministic.
#include "cnr_gemm_ws.h"

namespace cnr {

// Round 6: the MFMA SHAPE.  v_mfma_f32_16x16x32_f16 is 12 % cheaper per FLOP than v_mfma_f32_32x32x16_f16 on random operands and holds the clock 15 % higher
// (profiles/r06_mfma_shapes.txt: 1950 against 1705 TFLOP/s with the matrix pipe alone at the board's power limit) -- and this kernel's time is its joules.
// FD_MFMA16 = 1: product and weight gradient on 16 x 16 x 32 blocks (same LDS planes, same register budget: 128 weight registers, 16 / 128 accumulators);
// 0: the 32 x 32 x 16 form (A/B builds).
#ifndef FD_MFMA16
#define FD_MFMA16 1
#endif
typedef float fd_f32x4 __attribute__((ext_vector_type(4)));
constexpr int FD_TP = 32;                        // points per tile
constexpr int FD_ALD = 256 * 2 + 16;             // bytes per LDS row of one S plane (+16: conflict-free ds_read_b128 of the product fragments)
constexpr int FD_APLANE = FD_TP * FD_ALD;
constexpr int FD_ABUF = 2 * FD_APLANE + 256;     // two planes + rs[32] (1 / row scale) + ss[32] (row scale; 0 / NaN: zero / non-finite row)
constexpr int FD_YLD = 128 * 2 + 16;             // bytes per POINT row of one Ep' plane (round 6: row-major [point][column], see fd_tr8): 128 columns x 2 B + 16
constexpr int FD_YPLANE = FD_TP * FD_YLD;
constexpr int FD_YBUF = 2 * FD_YPLANE;
constexpr int FD_TLD = 36;
constexpr int FD_TBYTES = 32 * FD_TLD * 4;       // accumulator transposition buffer of one P wave
constexpr int FD_OFF_Y = 3 * FD_ABUF;
constexpr int FD_OFF_T = FD_OFF_Y + 2 * FD_YBUF;
constexpr int FD_OFF_INFO = FD_OFF_T + 4 * FD_TBYTES;   // [3 tiles][4 D waves] min log2(ss * se) of the rows a wave staged
constexpr int FD_LDS = FD_OFF_INFO + 64;
static_assert(FD_LDS <= 160 * 1024, "LDS budget of one CU");
constexpr int FD_GBIG = 0x3f000000;              // "no point with two non-zero rows yet"

// Round 6: BOTH operands of the weight-gradient MFMAs come out of row-major [point][column] f16 planes by the LDS transpose read
// (ds_read_b64_tr_b16: a 16-lane group reads a [4 points][16 columns] block, 8 contiguous bytes per lane, and every lane receives one column of
// it) -- S' from the planes the product uses anyway (16 LDS reads per wave and tile instead of 64 two-byte reads), Ep' from planes the product
// waves now write with ONE 8-byte store per plane and 4 columns (8 LDS stores per lane and tile instead of 32 two-byte transposed ones, no
// rotation of the column order against bank conflicts: same-box ablation before the change, backward pass 14.65 ms -> 14.14 ms with those
// stores removed).  The four rows of a block are points 4 apart: both row strides (528 B, 272 B) are 4 banks mod 64, so rows 4 points apart
// sit 16 banks apart and the 2 x 4 x 4 eight-byte pieces of a 32-lane half cover the 64 banks exactly once.  Position q = 4 h + r of k group
// kg in k16 block kb therefore holds point 16 kb + 4 r + 2 kg + h -- in both operands, which is all the MFMA needs.
// 2^G / sx for a power-of-two sx > 0 by exponent arithmetic (0 stays 0, NaN stays NaN, underflow flushes to 0)
__device__ __forceinline__ float fd_yscale(float sx, int G) {
  const unsigned bits = __float_as_uint(sx);
  const int field = G - (int)((bits >> 23) & 0xff) + 254;
  const float r = __uint_as_float((unsigned)(field < 1 ? 0 : (field > 254 ? 254 : field)) << 23);
  return sx > 0.0f ? (field < 1 ? 0.0f : r) : sx;
}

// the epilogue-side operand of 4 columns from the epilogue's side inputs (see DwFuse in cnr_views.h)
template <int EK>
__device__ __forceinline__ f4 fd_ep4(const Epi& e, const EpiRaw4& raw) {
  f4 r;
  if constexpr (EK == EK_RELU_MASK || EK == EK_SPLIT) {
    r = raw.a;
  } else if constexpr (EK == EK_VBACK) {
    r.x = softplus100(raw.a.x); r.y = softplus100(raw.a.y); r.z = softplus100(raw.a.z); r.w = softplus100(raw.a.w);
  } else {
    r.x = softplus100_d1(raw.a.x) * (raw.b.x * e.vscale); r.y = softplus100_d1(raw.a.y) * (raw.b.y * e.vscale);
    r.z = softplus100_d1(raw.a.z) * (raw.b.z * e.vscale); r.w = softplus100_d1(raw.a.w) * (raw.b.w * e.vscale);
  }
  return r;
}


// DP: the P waves keep the epilogue side inputs of TWO tiles in flight (tile i + 2 is requested while tile i is finished);
// DD: the D waves keep two input tiles in flight (tile i + 3 is requested when tile i + 1 has been converted).
// NKB: k16 blocks of the layer product (16; 14 for the 217-wide SDF layer in front of the skip connection, whose pad columns are zero)
// XR: DwFuse::xrow_mode (0 none; 1 column sums of the o2 output of an EK_SWEEP launch; 2 EK_VBACK launch with k_extra: rank-one update of the
// product by A column 256 and the row of that column's products with Ep) -- the sdf row of the 257-wide top SDF layer, see cnr_plan.cpp
// TAILF: the epilogue keeps its tail fill (sweep launch of the layer below a skip connection) and runs the general 16-byte epilogue code
// SPLITF: the value-backward epilogue keeps its split point (layer fed by a skip connection; see fdw_shape_ok), general 16-byte epilogue code
template <int EK, bool DP, bool DD, int NKB = 16, int XR = 0, bool TAILF = false, bool SPLITF = false>
__global__ __launch_bounds__(512, 1) void layer_dw_kernel(const LayerGemm g_in, const DwFuse f, int tiles_per_range, int dbg_in) {
  const int dbg = CNR_ABLATION(dbg_in);   // (CNR_FDW_DBG of the tuning build: parts of the kernel switched off; the constant 0 in the product build)
  LayerGemm g = g_in;
  g.A.kind = VK_DIRECT; g.E.kind = EK;
  if (!SPLITF) g.E.split = 1 << 30;
  if (!TAILF) { g.E.tail_src = nullptr; g.E.tail_n = 0; }
  constexpr bool GENF = TAILF || SPLITF;
  // side inputs of 4 columns: the split form needs z in every column (beyond the split point it is the epilogue-side operand itself)
  auto fetch_side = [&](long row, int col) {
    if constexpr (SPLITF) {
      EpiRaw4 r;
      const f4 zero = {0.f, 0.f, 0.f, 0.f};
      r.a = *reinterpret_cast<const f4*>(g.E.z + row * g.E.ldz + col);
      r.b = col < g.E.split ? *reinterpret_cast<const f4*>(g.E.o1 + row * g.E.ld1 + col) : zero;
      return r;
    } else if constexpr (EK == EK_SPLIT) {   // no epilogue side input: only the epilogue-side operand (aux)
      EpiRaw4 r;
      r.a = *reinterpret_cast<const f4*>(g.E.aux + row * g.E.ldaux + col);
      r.b = r.a;
      return r;
    } else {
      return epi_fetch4_sel<EK, GENF>(g.E, row, col);
    }
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // blocks b and b + 8 (same XCD under the observed round-robin placement: a speed matter only) are the two column halves of one range
  const int b = blockIdx.x;
  const int half = (b >> 3) & 1, range = (b >> 4) * 8 + (b & 7);
  const long ntiles = g.P / FD_TP;
  const long t0 = (long)range * tiles_per_range;
  const int n = t0 < ntiles ? (int)((ntiles - t0) < tiles_per_range ? (ntiles - t0) : tiles_per_range) : 0;
  int* info = reinterpret_cast<int*>(smem + FD_OFF_INFO);
  float* out = f.partial + (long)range * f.Npad * f.ldk;
  const bool isP = wave < 4;
  const int nlast = n > 0 ? n - 1 : 0;
  // f.rev: this launch walks every range downwards (the loop counter i maps to tile t0 + nlast - i): consecutive launches of a backward chain
  // alternate, so a launch starts on the rows its producer wrote last (still in L2 / Infinity Cache).  Fixed per launch site: deterministic
  const int rev = f.rev;
#define FD_TILE(i_) (t0 + (rev ? nlast - (i_) : (i_)))

  if (isP) {
    // ================================================================ P waves
    const int c0 = half * 128 + wave * 32;               // first output column of this wave
#if FD_MFMA16
    // weights of this wave's 2 x 16 output columns as B fragments of the 16 x 16 x 32 MFMA: lane (n = lane & 15, kg = lane >> 4) holds W[c0 + 16 cb + n][32 kb + 8 kg ..]
    constexpr int NKB2 = NKB / 2;
    f16x8 w1[NKB2][2], w2[NKB2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const unsigned short* wp = g.Wp + (long)(c0 + 16 * cb + (lane & 15)) * g.ldw + (lane >> 4) * 8;
#pragma unroll
      for (int kb = 0; kb < NKB2; ++kb) {
        w1[kb][cb] = *reinterpret_cast<const f16x8*>(wp + kb * 32);
        w2[kb][cb] = *reinterpret_cast<const f16x8*>(wp + g.wp_stride + kb * 32);
      }
    }
#else
    f16x8 w1[NKB], w2[NKB];
    {
      const unsigned short* wp = g.Wp + (long)(c0 + (lane & 31)) * g.ldw + (lane >> 5) * 8;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        w1[kb] = *reinterpret_cast<const f16x8*>(wp + kb * 16);
        w2[kb] = *reinterpret_cast<const f16x8*>(wp + g.wp_stride + kb * 16);
      }
    }
#endif
    const int ecol = c0 + (lane & 7) * 4;
    const f4 wsc = *reinterpret_cast<const f4*>(g.wscale + ecol);
    const f4 bias4 = epi_bias4(g.E, ecol);
    float* T = reinterpret_cast<float*>(smem + FD_OFF_T + wave * FD_TBYTES);
    int G = FD_GBIG;
    f4 wr1 = {0.f, 0.f, 0.f, 0.f}, xacc = {0.f, 0.f, 0.f, 0.f};   // XR: column 256 of this lane's 4 weight rows; the extra row's sums
    float xb = 0.0f;
    if constexpr (XR == 2) {
      wr1.x = g.W[(long)(ecol + 0) * g.ldw + 256]; wr1.y = g.W[(long)(ecol + 1) * g.ldw + 256];
      wr1.z = g.W[(long)(ecol + 2) * g.ldw + 256]; wr1.w = g.W[(long)(ecol + 3) * g.ldw + 256];
    }
    // one tile: product, epilogue, Ep' tile; `ern` holds this tile's side inputs and receives those of tile i + ahead (clamped to the range)
    auto p_tile = [&](const int i, const int ab, EpiRaw4 (&ern)[4], const int ahead) {
      const long t = FD_TILE(i);
      const unsigned char* B = smem + ab * FD_ABUF;
      {
        const int* qi = info + ab * 4;
        int m = qi[0]; m = qi[1] < m ? qi[1] : m; m = qi[2] < m ? qi[2] : m; m = qi[3] < m ? qi[3] : m;
        if (m < FD_GBIG && m + 1 < G) G = m + 1;
      }
      float zq[4] = {0.f, 0.f, 0.f, 0.f};                // XR == 2: A[row][256] of this lane's 4 rows (requested before the product, used after it)
      if constexpr (XR == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) zq[q] = g.A.a[(t * FD_TP + (lane >> 3) + 8 * q) * g.A.lda + 256];
      }
#if FD_MFMA16
      fd_f32x4 acc[2][2];   // [row block of 16 points][column block of 16 columns]
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[rb][cb][j] = 0.0f;
      const unsigned char* Ab = B + (lane & 15) * FD_ALD + (lane >> 4) * 16;   // A fragment: S'[16 rb + (lane & 15)][32 kb + 8 (lane >> 4) ..]
      if (!(dbg & 4))
#pragma unroll
      for (int kb = 0; kb < NKB2; ++kb) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + rb * 16 * FD_ALD + kb * 64);
          const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + rb * 16 * FD_ALD + FD_APLANE + kb * 64);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            fd_f32x4 c = acc[rb][cb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w2[kb][cb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, w1[kb][cb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w1[kb][cb], c, 0, 0, 0);
            acc[rb][cb] = c;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* rs = reinterpret_cast<const float*>(B + 2 * FD_APLANE);
      const float* ssr = rs + 32;
      {
        const int q4 = lane >> 4, cl = lane & 15;   // result block: rows 4 q4 + r, column cl
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(16 * rb + 4 * q4 + r) * FD_TLD + 16 * cb + cl] = acc[rb][cb][r];
      }
#else
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
      const unsigned char* Ab = B + (lane & 31) * FD_ALD + (lane >> 5) * 16;
      if (!(dbg & 4))
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32);
        const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + FD_APLANE + kb * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* rs = reinterpret_cast<const float*>(B + 2 * FD_APLANE);
      const float* ssr = rs + 32;
      const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * FD_TLD + cl] = acc[r];
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const long tn = FD_TILE(i + ahead < nlast ? i + ahead : nlast);
      unsigned char* Yb = smem + FD_OFF_Y + (i & 1) * FD_YBUF;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = (lane >> 3) + 8 * q, cc = (lane & 7) * 4;
        const long row = t * FD_TP + rr;
        const float rsc = rs[rr];
        f4 v = *reinterpret_cast<const f4*>(T + rr * FD_TLD + cc);
        v.x *= rsc * wsc.x; v.y *= rsc * wsc.y; v.z *= rsc * wsc.z; v.w *= rsc * wsc.w;
        f4 ep = fd_ep4<EK>(g.E, ern[q]);
        if constexpr (SPLITF) {   // [softplus(z) | z] * vscale: the layer's forward input as its view forms it
          const f4 zz = ern[q].a;
          const int sp = g.E.split - ecol;
          ep.x = (sp > 0 ? ep.x : zz.x) * g.E.vscale; ep.y = (sp > 1 ? ep.y : zz.y) * g.E.vscale;
          ep.z = (sp > 2 ? ep.z : zz.z) * g.E.vscale; ep.w = (sp > 3 ? ep.w : zz.w) * g.E.vscale;
        }
        if constexpr (XR == 2) {
          const float zs = zq[q];
          v.x = fmaf(zs, wr1.x, v.x); v.y = fmaf(zs, wr1.y, v.y); v.z = fmaf(zs, wr1.z, v.z); v.w = fmaf(zs, wr1.w, v.w);
          xacc.x = fmaf(zs, ep.x, xacc.x); xacc.y = fmaf(zs, ep.y, xacc.y); xacc.z = fmaf(zs, ep.z, xacc.z); xacc.w = fmaf(zs, ep.w, xacc.w);
          xb += zs;
        }
        if constexpr (XR == 1) {   // the o2 values of the sweep epilogue (same expression as in epi_finish4_plain)
          xacc.x += softplus100_d1(ern[q].a.x) * v.x; xacc.y += softplus100_d1(ern[q].a.y) * v.y;
          xacc.z += softplus100_d1(ern[q].a.z) * v.z; xacc.w += softplus100_d1(ern[q].a.w) * v.w;
        }
        epi_finish4_sel<EK, GENF>(g.E, row, ecol, v, bias4, ern[q]);
        ern[q] = fetch_side(tn * FD_TP + rr, ecol);
        const float ys = fd_yscale(ssr[rr], G);
        ep.x *= ys; ep.y *= ys; ep.z *= ys; ep.w *= ys;
        unsigned char* yrow = Yb + rr * FD_YLD + (wave * 32 + cc) * 2;
        if (!(dbg & 2)) {
          ws_f16x4 h1, h2;
          h1[0] = (_Float16)ep.x; h1[1] = (_Float16)ep.y; h1[2] = (_Float16)ep.z; h1[3] = (_Float16)ep.w;
          h2[0] = (_Float16)(ep.x - (float)h1[0]); h2[1] = (_Float16)(ep.y - (float)h1[1]);
          h2[2] = (_Float16)(ep.z - (float)h1[2]); h2[3] = (_Float16)(ep.w - (float)h1[3]);
          *reinterpret_cast<ws_f16x4*>(yrow) = h1;
          *reinterpret_cast<ws_f16x4*>(yrow + FD_YPLANE) = h2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    EpiRaw4 ernA[4], ernB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ernA[q] = fetch_side(FD_TILE(0) * FD_TP + (lane >> 3) + 8 * q, ecol);   // (a range past the end is clamped to tile 0 of the launch: loaded, never used)
      if (DP) ernB[q] = fetch_side(FD_TILE(1 < nlast ? 1 : nlast) * FD_TP + (lane >> 3) + 8 * q, ecol);
    }
    cnr_lds_barrier();   // tile 0 staged
    int ab = 0;
    if (!DP) {
      for (int i = 0; i <= n; ++i) {
        if (i < n) p_tile(i, ab, ernA, 1);
        ab = ab == 2 ? 0 : ab + 1;
        cnr_lds_barrier();
      }
    } else {
      for (int i = 0; i <= n; i += 2) {
        if (i < n) p_tile(i, ab, ernA, 2);
        ab = ab == 2 ? 0 : ab + 1;
        cnr_lds_barrier();
        if (i + 1 <= n) {
          if (i + 1 < n) p_tile(i + 1, ab, ernB, 2);
          ab = ab == 2 ? 0 : ab + 1;
          cnr_lds_barrier();
        }
      }
    }
    if constexpr (XR != 0) {
      // the extra row of this (range, half): fold the 8 row lanes (lane >> 3) in a fixed order; lanes 0..7 then hold this wave's 32 columns
#pragma unroll
      for (int d = 8; d <= 32; d <<= 1) {
        xacc.x += __shfl_xor(xacc.x, d); xacc.y += __shfl_xor(xacc.y, d); xacc.z += __shfl_xor(xacc.z, d); xacc.w += __shfl_xor(xacc.w, d);
        xb += __shfl_xor(xb, d);
      }
      if (lane < 8) {
        float* xr = f.xrow + (long)range * f.xrow_stride + ecol;
        f4 o;
        o.x = xacc.x * f.xrow_scale; o.y = xacc.y * f.xrow_scale; o.z = xacc.z * f.xrow_scale; o.w = xacc.w * f.xrow_scale;
        if constexpr (XR == 2) { const f4 old = *reinterpret_cast<const f4*>(xr); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<f4*>(xr) = o;
      }
      if (XR == 2 && f.xbias != nullptr && half == 0 && wave == 0 && lane == 0) f.xbias[(long)range * f.xbias_stride] = xb;   // (every column lane of the wave saw the same rows)
    }
  } else {
    // ================================================================ D waves
    // the staging waves go first at the issue arbiter: their global loads are what everything else waits for (measured: -0.58 ms per step
    // against equal priorities; the product waves first: no change).  CNR_FDW_DBG bits 3 / 4: priority 0 / 3 instead (tuning aid)
    if (dbg & 8) __builtin_amdgcn_s_setprio(0); else if (dbg & 16) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
    const int wd = wave - 4, dt = tid - 256;
    const int srow = dt >> 4, scol = (dt & 15) * 4;
#if FD_MFMA16
    const int m = lane & 15, kg = lane >> 4;   // result blocks: column m, rows 4 kg + r
    fd_f32x4 acc[4][8];                        // [block of 16 S columns (rows of dW)][block of 16 Ep columns]: 128 registers
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
#else
    const int m = lane & 31, kg = lane >> 5;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#endif
    struct RawTile { f4 r[2][4]; float se[2]; };
    f4 cs[4];
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) cs[q] = z4;
    const bool want_cs = f.colsum != nullptr && half == 0;
    const float ascale = g.A.scale;

    auto d_fetch = [&](int i, RawTile& rt) {
      const long t = FD_TILE(i < nlast ? i : nlast);   // (clamped: the redundant request at the end of a range is never converted)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const long row = t * FD_TP + srow + 16 * p;
        const float* src = g.A.a + row * g.A.lda + scol;
#pragma unroll
        for (int q = 0; q < 4; ++q) rt.r[p][q] = *reinterpret_cast<const f4*>(src + 64 * q);
        rt.se[p] = f.se[row];
      }
    };
    auto d_put = [&](int i, int ab, const RawTile& rt) {
      unsigned char* B = smem + ab * FD_ABUF;
      float* rs = reinterpret_cast<float*>(B + 2 * FD_APLANE);
      int qmin = FD_GBIG;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        f4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = rt.r[p][q];
          if (ascale != 1.0f) { v[q].x *= ascale; v[q].y *= ascale; v[q].z *= ascale; v[q].w *= ascale; }
        }
        float mx = fmaxf(fmaxf(ws_absmax4(v[0]), ws_absmax4(v[1])), fmaxf(ws_absmax4(v[2]), ws_absmax4(v[3])));
        mx = cnr_max16(mx);
        const bool valid = mx > 0.0f && mx < 3.0e38f;
        float sc = 1.0f;
        if (valid) { int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100; sc = ldexpf(1.0f, 14 - e_); }
        const int row_l = srow + 16 * p;
        unsigned char* dst = B + row_l * FD_ALD + scol * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) ws_put4(v[q], sc, dst + 128 * q, FD_APLANE);
        if (want_cs) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { cs[q].x += v[q].x; cs[q].y += v[q].y; cs[q].z += v[q].z; cs[q].w += v[q].w; }
        }
        const float ssv = valid ? sc : (mx == 0.0f ? 0.0f : __builtin_nanf(""));
        if ((dt & 15) == 0) {
          rs[row_l] = cnr_pow2_rcp(sc);
          rs[32 + row_l] = ssv;
          if (g.rs_out && half == 0) g.rs_out[FD_TILE(i) * FD_TP + row_l] = ssv;
        }
        const float se = rt.se[p];
        if (valid && se > 0.0f) {
          const int e = (int)((__float_as_uint(sc) >> 23) & 0xff) + (int)((__float_as_uint(se) >> 23) & 0xff) - 254;
          qmin = e < qmin ? e : qmin;
        }
      }
      qmin = cnr_pair32_min(cnr_pair16_min(qmin));   // (rows 16 apart, then the two halves: VALU lane swaps, not LDS-crossbar shuffles)
      if (lane == 0) info[ab * 4 + wd] = qmin;
    };
    auto d_dw = [&](int i, int ab) {
      const unsigned char* B = smem + ab * FD_ABUF;
      const unsigned char* Yb = smem + FD_OFF_Y + (i & 1) * FD_YBUF;
#if FD_MFMA16
      {
        // ONE k32 step covers the tile's 32 points.  This lane's 8-byte piece of a [4 points][16 columns] block: k group kg = lane >> 4 holds points
        // 16 (kg >> 1) + 2 (kg & 1) + 4 r + h (the two 16-lane groups of a 32-lane half are then 8 banks apart, the four rows 16 banks: conflict-free)
        const int prow = 16 * (kg >> 1) + 2 * (kg & 1) + 4 * ((lane & 15) >> 2), pcol = (lane & 3) * 4;
        f16x8 a[4][2];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) a[sb][pl] = ws_tr8(B + pl * FD_APLANE + prow * FD_ALD + (wd * 64 + sb * 16 + pcol) * 2, FD_ALD);
#pragma unroll
        for (int eb = 0; eb < 8; ++eb) {
          const unsigned char* ysrc = Yb + prow * FD_YLD + (eb * 16 + pcol) * 2;
          const f16x8 b1 = ws_tr8(ysrc, FD_YLD);
          const f16x8 b2 = ws_tr8(ysrc + FD_YPLANE, FD_YLD);
#pragma unroll
          for (int sb = 0; sb < 4; ++sb) {
            fd_f32x4 c = acc[sb][eb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[sb][0], b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[sb][1], b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[sb][0], b1, c, 0, 0, 0);
            acc[sb][eb] = c;
          }
        }
      }
#else
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        // this lane's 8-byte piece of a [4 points][16 columns] block: point row kb * 16 + 2 kg + 4 r (h = 1: one row below), 4 columns at pcol
        const int prow = kb * 16 + kg * 2 + 4 * ((lane & 15) >> 2), pcol = (m & 16) + (lane & 3) * 4;
        f16x8 a[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            a[it][pl] = ws_tr8(B + pl * FD_APLANE + prow * FD_ALD + (wd * 64 + it * 32 + pcol) * 2, FD_ALD);
          }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const unsigned char* ysrc = Yb + prow * FD_YLD + (jt * 32 + pcol) * 2;
          const f16x8 b1 = ws_tr8(ysrc, FD_YLD);
          const f16x8 b2 = ws_tr8(ysrc + FD_YPLANE, FD_YLD);
          f32x16 c0 = acc[0][jt], c1 = acc[1][jt];
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], b2, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], b2, c1, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1], b1, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1], b1, c1, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], b1, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], b1, c1, 0, 0, 0);
          acc[0][jt] = c0; acc[1][jt] = c1;
        }
      }
#endif
    };
    int Gd = FD_GBIG;
    // weight-gradient contribution of tile i (buffer abp): fold the tile's exponent into the running one first
    auto d_acc = [&](int i, int abp) {
      const int* qi = info + abp * 4;
      int mq = qi[0]; mq = qi[1] < mq ? qi[1] : mq; mq = qi[2] < mq ? qi[2] : mq; mq = qi[3] < mq ? qi[3] : mq;
      mq = __builtin_amdgcn_readfirstlane(mq);
      if (mq < FD_GBIG && mq + 1 < Gd) {
        if (Gd < FD_GBIG) {   // exact power-of-two rescale of what has been accumulated under the old exponent
          const int dlt = mq + 1 - Gd;
          const float u1 = ldexpf(1.0f, dlt / 2), u2 = ldexpf(1.0f, dlt - dlt / 2);
#if FD_MFMA16
#pragma unroll
          for (int sb = 0; sb < 4; ++sb)
#pragma unroll
            for (int eb = 0; eb < 8; ++eb)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[sb][eb][r] = acc[sb][eb][r] * u1 * u2;
#else
#pragma unroll
          for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[it][jt][r] = acc[it][jt][r] * u1 * u2;
#endif
        }
        Gd = mq + 1;
      }
      if (!(dbg & 1)) d_dw(i, abp);
    };

    RawTile ta, tb;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { ta.r[p][q] = z4; tb.r[p][q] = z4; }
      ta.se[p] = 0.f; tb.se[p] = 0.f;
    }
    int ab = 0;   // buffer of tile i
    if (!DD) {
      if (n > 0) { d_fetch(0, ta); d_put(0, 0, ta); }
      if (n > 1) d_fetch(1, ta);
      cnr_lds_barrier();   // tile 0 staged
      for (int i = 0; i <= n; ++i) {
        const int abn = ab == 2 ? 0 : ab + 1, abp = ab == 0 ? 2 : ab - 1;
        if (i + 1 < n) d_put(i + 1, abn, ta);
        if (i + 2 < n) d_fetch(i + 2, ta);
        if (i >= 1) d_acc(i - 1, abp);
        ab = abn;
        cnr_lds_barrier();
      }
    } else {
      // tile j travels in set a for even j, in set b for odd j; a set is re-requested (tile j + 2) right after its conversion
      if (n > 0) { d_fetch(0, ta); d_put(0, 0, ta); }
      d_fetch(1, tb);
      d_fetch(2, ta);
      cnr_lds_barrier();   // tile 0 staged
      for (int i = 0; i <= n; i += 2) {
        {
          const int abn = ab == 2 ? 0 : ab + 1, abp = ab == 0 ? 2 : ab - 1;
          if (i + 1 < n) d_put(i + 1, abn, tb);
          d_fetch(i + 3, tb);
          if (i >= 1) d_acc(i - 1, abp);
          ab = abn;
          cnr_lds_barrier();
        }
        if (i + 1 <= n) {
          const int abn = ab == 2 ? 0 : ab + 1, abp = ab == 0 ? 2 : ab - 1;
          if (i + 2 < n) d_put(i + 2, abn, ta);
          d_fetch(i + 4, ta);
          d_acc(i, abp);
          ab = abn;
          cnr_lds_barrier();
        }
      }
    }

    // ---- partial sums of this (range, half): undo 2^G in two exact steps; the transposed form goes through a per-wave LDS tile so that
    // both forms store 128-byte row pieces
    if (Gd >= FD_GBIG) Gd = 0;
    const float u1 = ldexpf(1.0f, -(Gd / 2)), u2 = ldexpf(1.0f, -(Gd - Gd / 2));
    float* X = reinterpret_cast<float*>(smem + wd * (32 * 33 * 4));
#if FD_MFMA16
    // a lane holds dW[s = 16 sb + 4 kg + r][e = 16 eb + m] of each block.  Natural form: 64-byte row pieces; transposed form: 32 x 32 regions (2 x 2 blocks)
    // through the per-wave LDS tile so that both forms store whole 128-byte row pieces where they can
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int s0 = wd * 64 + it * 32, e0 = half * 128 + jt * 32;
        if (!f.transposed) {
#pragma unroll
          for (int sbl = 0; sbl < 2; ++sbl)
#pragma unroll
            for (int ebl = 0; ebl < 2; ++ebl)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int sidx = s0 + 16 * sbl + 4 * kg + r;
                if (sidx < f.Npad) out[(long)sidx * f.ldk + e0 + 16 * ebl + m] = acc[2 * it + sbl][2 * jt + ebl][r] * u1 * u2;
              }
        } else {
#pragma unroll
          for (int sbl = 0; sbl < 2; ++sbl)
#pragma unroll
            for (int ebl = 0; ebl < 2; ++ebl)
#pragma unroll
              for (int r = 0; r < 4; ++r) X[(16 * sbl + 4 * kg + r) * 33 + 16 * ebl + m] = acc[2 * it + sbl][2 * jt + ebl][r] * u1 * u2;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          {
            const int m32 = lane & 31, k2 = lane >> 5;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int e = 2 * r + k2;
              if (e0 + e < f.Npad) out[(long)(e0 + e) * f.ldk + s0 + m32] = X[m32 * 33 + e];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
#else
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int s0 = wd * 64 + it * 32, e0 = half * 128 + jt * 32;
        if (!f.transposed) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (s < f.Npad) out[(long)s * f.ldk + e0 + m] = acc[it][jt][r] * u1 * u2;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) X[((r & 3) + 8 * (r >> 2) + 4 * kg) * 33 + m] = acc[it][jt][r] * u1 * u2;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int e = 2 * r + kg;
            if (e0 + e < f.Npad) out[(long)(e0 + e) * f.ldk + s0 + m] = X[m * 33 + e];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
#endif
    if (want_cs) {   // bias gradient: the 16 row owners of a column group are folded in a fixed order below
      float* C = reinterpret_cast<float*>(smem + 4 * (32 * 33 * 4));
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(C + srow * 256 + scol + 64 * q) = cs[q];
    }
  }
  if (f.colsum != nullptr && half == 0) {
    cnr_lds_barrier();
    if (!isP) {
      const int dt = tid - 256;
      const float* C = reinterpret_cast<const float*>(smem + 4 * (32 * 33 * 4));
      float sum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += C[r * 256 + dt];
      if (dt < f.Npad) f.colsum[(long)range * f.Npad + dt] = sum;
    }
  }
}

#undef FD_TILE

template <int EK, bool DP, bool DD, int NKB = 16, int XR = 0, bool TAILF = false, bool SPLITF = false>
static void launch_fdw_v(const LayerGemm& g, const DwFuse& f, cnr_stream s) {
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_dw_kernel<EK, DP, DD, NKB, XR, TAILF, SPLITF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long ntiles = g.P / FD_TP;
  const int tpr = (int)((ntiles + f.nslots - 1) / f.nslots);
  TimingScope ts_("layer_dw", 0, 200 + EK, g.P, 256, g.K, 2, s, fdw_bytes(g, f));   // pairs = 2: the layer product + the weight-gradient product
  const int dbg = debug_flags().fdw_dbg;   // ablation word of the tuning build (timing experiments only: results are wrong); 0 in the product build
  hipLaunchKernelGGL((layer_dw_kernel<EK, DP, DD, NKB, XR, TAILF, SPLITF>), dim3(2 * f.nslots), dim3(512), FD_LDS, s, g, f, tpr, dbg);
}
template <int EK>
static void launch_fdw(const LayerGemm& g, const DwFuse& f, cnr_stream s) {
#ifdef CNR_TUNING
  switch (debug_flags().fdw_deep & 3) {   // tuning aid: bit 0 = DP, bit 1 = DD (deeper prefetch: 1-5 % slower, spills; instantiated in the tuning build only)
    case 0: launch_fdw_v<EK, false, false>(g, f, s); break;
    case 1: launch_fdw_v<EK, true, false>(g, f, s); break;
    case 2: launch_fdw_v<EK, false, true>(g, f, s); break;
    default: launch_fdw_v<EK, true, true>(g, f, s); break;
  }
#else
  launch_fdw_v<EK, false, false>(g, f, s);
#endif
}

bool be_fdw_enabled() {
  return !debug_flags().no_fdw;   // debugging aid: separate layer and weight-gradient launches everywhere
}
bool be_fdw_xrow() {   // the fused launches can carry DwFuse::xrow_mode (not when their slots are filled by the separate kernels)
  return be_fdw_enabled() && !debug_flags().fdw_split;
}

void be_layer_dw_gemm(const LayerGemm& g, const DwGemm& d, const DwFuse& f, cnr_stream s) {
  const bool split_env = debug_flags().fdw_split;   // debugging aid: the same slots filled by the two separate kernels
  const bool slots_ok = f.se != nullptr && f.nslots >= 8 && (f.nslots & 7) == 0 && f.nslots <= kFdwSlots;
  if (split_env || !slots_ok || !fdw_shape_ok(g) || f.Npad < 224 || f.Npad > 288 || f.ldk < 256 || f.ldk > 320) {
    if (f.xrow_mode != 0 || g.k_extra != 0) { if (g_first_error == hipSuccess) { g_first_error = hipErrorInvalidValue; g_first_error_where = "layer_dw: the extra-row form needs the fused launch (callers test be_fdw_xrow() and the shape)"; } return; }
    be_dw_gemm(d, s);
    be_layer_gemm(g, s);
    return;
  }
  if ((f.xrow_mode != 0 || g.k_extra != 0) && !((f.xrow_mode == 2 && g.E.kind == EK_VBACK && g.k_extra == 1 && g.K == 256) || (f.xrow_mode == 1 && g.E.kind == EK_SWEEP && g.k_extra == 0 && g.K > 240))) {
    if (g_first_error == hipSuccess) { g_first_error = hipErrorInvalidValue; g_first_error_where = "layer_dw: unsupported extra-row request"; }
    return;
  }
  // the fused launch covers output columns [0, 256) and the 256 x 256 main tile of the weight gradient
  if (f.xrow_mode == 2 && g.E.kind == EK_VBACK && g.k_extra == 1 && g.K == 256) launch_fdw_v<EK_VBACK, false, false, 16, 2>(g, f, s);
  else if (f.xrow_mode == 1 && g.E.kind == EK_SWEEP && g.K > 240) launch_fdw_v<EK_SWEEP, false, false, 16, 1>(g, f, s);
  else if (g.E.kind == EK_SWEEP && g.E.tail_src != nullptr) launch_fdw_v<EK_SWEEP, false, false, 16, 0, true>(g, f, s);
  else if (g.E.kind == EK_VBACK && g.E.split < 256) launch_fdw_v<EK_VBACK, false, false, 16, 0, false, true>(g, f, s);
  else if (g.K <= 224) launch_fdw_v<EK_VBACK, false, false, 14>(g, f, s);
  else switch (g.E.kind) {
    case EK_RELU_MASK: launch_fdw<EK_RELU_MASK>(g, f, s); break;
    case EK_VBACK: launch_fdw<EK_VBACK>(g, f, s); break;
    case EK_SPLIT: launch_fdw_v<EK_SPLIT, false, false>(g, f, s); break;
    default: launch_fdw<EK_SWEEP>(g, f, s); break;
  }
  CNR_LAUNCH_CHECK("layer_dw");
  // a layer with a few more input columns (relight y-layer: 256 hidden + rgb): their cotangent columns by the narrow layer kernel, the
  // weight-gradient strip beyond column 256 by the strip kernels into the same slots
  if (g.N > 256) { LayerGemm tail = g; tail.first_col = 256; tail.rs_out = nullptr; be_layer_gemm(tail, s); }
  if (d.K > 256) { DwGemm strips = d; strips.skip_main = true; strips.colsum = nullptr; be_dw_gemm(strips, s); }
}

}  // namespace cnr
