// Weight-stationary layer GEMM: kernel template and launcher (included by cnr_gemm_ws_a.hip / cnr_gemm_ws_b.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"

namespace cnr {

// ================================================================================================
// weight-stationary layer GEMM (K <= 256, up to 256 output columns per launch)
//
// The whole layer lives in the REGISTER FILE of one CU: 8 waves x 32 output columns; each wave keeps its slice of W as two f16
// planes in MFMA B-operand layout (2 planes x 16 k-blocks x 4 VGPRs = 128 VGPRs).  Points stream through in 32-row tiles: a tile is
// fetched once, gets the fused prologue (cnr_views.h), is scaled by an exact power of two per row, split into f16 hi + lo
// (11 + 11 significand bits) on its way into LDS and read by all 8 waves.  Each product is three v_mfma_f32_32x32x16_f16
// (a1 w1 + a1 w2 + a2 w1; the dropped a2 w2 is < 2^-22), fp32 accumulation, and the epilogue undoes the row / column scales
// (exact).  Measured (tools/probes/ws_probe.hip): max error 8.3e-7 vs float64 where the FP32 FMA chain of v_mfma_f32_32x32x2_f32 has
// 1.1e-6 -- also with rows spanning 12 orders of magnitude -- at 3 x 32 = 96 MFMA cycles per k16 block instead of 8 x 64 = 512.
// No weight traffic after the prologue, 16 accumulator registers, one barrier per 32 points.
// ================================================================================================
#ifndef WS_MFMA16
#define WS_MFMA16 1   // the stream form of the layer kernel on v_mfma_f32_16x16x32_f16 (0: 32 x 32 x 16, A/B builds)
#endif
#ifndef WS_GEN_MFMA16
// ... and the general form.  Built and parity-green on 16 x 16 x 32 too (then bit-identical to the stream form), but SLOWER there: 478 against 358 us for the
// one general launch of a step (the 217-wide layer of the forward gradient chain; profiles/r06_ab_general_mfma16.txt) -- its k blocks sit behind run-time
// guards, so nothing is scheduled across them.  It stays on 32 x 32 x 16: the two forms then agree to fp32 round-off, not to the bit
// (tests/test_hip_parity.py::test_stream_form_of_the_layer_kernel_matches_the_general_form).
#define WS_GEN_MFMA16 0
#endif
typedef float ws_f32x4 __attribute__((ext_vector_type(4)));
constexpr int WS_TP = 32;
constexpr int WS_THREADS = 512;
constexpr int WS_TLD = 36;

__device__ __forceinline__ float ws_dot4(const f4& a, const f4& b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

__device__ __forceinline__ void ws_put4(const f4& v, float sc, unsigned char* dst, int aplane) {
  f16x4 h1, h2;
  float x;
  x = v.x * sc; h1[0] = (_Float16)x; h2[0] = (_Float16)(x - (float)h1[0]);
  x = v.y * sc; h1[1] = (_Float16)x; h2[1] = (_Float16)(x - (float)h1[1]);
  x = v.z * sc; h1[2] = (_Float16)x; h2[2] = (_Float16)(x - (float)h1[2]);
  x = v.w * sc; h1[3] = (_Float16)x; h2[3] = (_Float16)(x - (float)h1[3]);
  *reinterpret_cast<f16x4*>(dst) = h1;
  *reinterpret_cast<f16x4*>(dst + aplane) = h2;
}

// VK / EK >= 0 pin the view / epilogue kind at compile time: the interpreted switches of cnr_views.h fold away and each
// instantiation only allocates the registers its own prologue and epilogue need (-1 = generic, interpreted at run time).
// PLAIN promises an epilogue without tail fill and without a split point (most launches): that code folds away as well.
// K17 admits a 17th k16 block (K up to 272: the layers whose input is a 256-wide hidden vector plus a few concatenated columns).
// FULLK promises K > (NKB - 1) * 16, i.e. every k16 block is live: the per-block guards fold away and the LDS fragment reads of the
// next block can be scheduled across the MFMAs of the current one.
template <int VK, int EK, bool PLAIN, bool K17 = false, bool FULLK = false>
__global__ __launch_bounds__(WS_THREADS, 1) void layer_gemm_ws_kernel(const LayerGemm g_in, int tiles_per_wg, int wrows, int rev) {
  constexpr int NKB = K17 ? 17 : 16;
  LayerGemm g = g_in;
  if (VK >= 0) g.A.kind = VK;
  if (EK >= 0) g.E.kind = EK;
  if (PLAIN) { g.E.tail_src = nullptr; g.E.tail_n = 0; g.E.split = 1 << 30; }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long Pn = g.P_dev ? (long)*g.P_dev : g.P;
  const long ntiles = (Pn + WS_TP - 1) / WS_TP;
  const long t0 = (long)blockIdx.x * tiles_per_wg;
  if (t0 >= ntiles) return;
  long t1 = t0 + tiles_per_wg;
  if (t1 > ntiles) t1 = ntiles;
  // rev: this workgroup walks its tile range downwards (WS_TILE maps the loop counter to the tile): consecutive launches of a
  // chain alternate, so a launch starts on the rows its producer wrote last (still in L2 / MALL)
  const long tflip = t0 + t1 - 1;
#define WS_TILE(t_) (rev ? tflip - (t_) : (t_))
  const int nkb = FULLK ? NKB : (g.K + 15) >> 4;   // 1..NKB k16 blocks
  const int kpad = nkb * 16;
  const int ald = kpad * 2 + 16;                // bytes per LDS row of one plane (+16: conflict-free ds_read_b128)
  const int aplane = WS_TP * ald;
  const int abuf = 2 * aplane + 128;            // two planes + 32 row scales
  float* T = reinterpret_cast<float*>(smem_b + 2 * abuf) + wave * (32 * WS_TLD);
  const int c0 = g.col0 + wave * 32;            // this wave's first output column
  const bool has_w = c0 < wrows;
  const int ncols_live = g.E.n_out + (g.E.tail_src ? g.E.tail_n : 0);

  // ---- resident weights (two f16 planes of this wave's 32 rows of W)
#if WS_GEN_MFMA16
  // v_mfma_f32_16x16x32_f16 like the stream form below (one arithmetic for every split-f16 layer product of the library): k32 blocks; when the
  // number of live k16 blocks is odd, the lanes that hold the upper 16 k of the last block (lane >> 4 >= 2) carry zeros in both operands
  constexpr int NKB2 = (NKB + 1) / 2;
  f16x8 w1[NKB2][2], w2[NKB2][2];
  const bool khi = (lane >> 4) >= 2;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const unsigned short* wp = g.Wp + (long)(has_w ? c0 + 16 * cb + (lane & 15) : 0) * g.ldw + (lane >> 4) * 8;
#pragma unroll
    for (int kb = 0; kb < NKB2; ++kb) {
      if (2 * kb < nkb) {
        w1[kb][cb] = zero8; w2[kb][cb] = zero8;
        if (!(khi && 2 * kb + 1 == nkb)) {
          w1[kb][cb] = *reinterpret_cast<const f16x8*>(wp + kb * 32);
          w2[kb][cb] = *reinterpret_cast<const f16x8*>(wp + g.wp_stride + kb * 32);
        }
      }
    }
  }
#else
  f16x8 w1[NKB], w2[NKB];
  {
    const unsigned short* wp = g.Wp + (long)(has_w ? c0 + (lane & 31) : 0) * g.ldw + (lane >> 5) * 8;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb < nkb) {
        w1[kb] = *reinterpret_cast<const f16x8*>(wp + kb * 16);
        w2[kb] = *reinterpret_cast<const f16x8*>(wp + g.wp_stride + kb * 16);
      }
    }
  }
#endif
  f4 wsc = {1.f, 1.f, 1.f, 1.f};               // inverse column scales of the 4 columns this lane finishes in the epilogue
  if (has_w) wsc = *reinterpret_cast<const f4*>(g.wscale + c0 + (lane & 7) * 4);
  const int ecol = c0 + (lane & 7) * 4;         // ... and their epilogue path / bias (fixed per lane for the whole launch)
  const bool efast = epi_fast4(g.E, ecol);
  f4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (efast) bias4 = epi_bias4(g.E, ecol);

  // ---- staging map: 16 threads per row, 4 consecutive columns each, up to 4 passes of 64 columns
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  const bool pv0 = scol < kpad, pv1 = 64 + scol < kpad, pv2 = 128 + scol < kpad, pv3 = 192 + scol < kpad;
  const bool pv4 = K17 && 256 + scol < kpad;   // 17th block: 4 of the 16 threads of a row
  // DEEP: a second staging register set keeps the activation tiles of two iterations in flight (+7 % on a plain layer,
  // tools/probes/ws_depth_probe.hip); only the instantiations with >= 20 spare VGPRs take it
  constexpr bool DEEP = PLAIN && !K17 && FULLK && (VK == VK_DIRECT || VK == VK_SOFTPLUS) && (EK == EK_STORE || EK == EK_RELU || EK == EK_SDF_TOP);
  Raw4 r0a, r1a, r2a, r3a, r4a, r0b, r1b, r2b, r3b, r4b;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  constexpr bool WS_DOT = (EK == EK_SDF_TOP || EK < 0);   // the row dot is compiled into the top SDF layer's instantiation (and the runtime-kind one)
  f4 dw0 = z4, dw1 = z4, dw2 = z4, dw3 = z4;      // row-dot weights of this thread's columns (LayerGemm::dot_w; K <= 256 there)
  float dot_b = 0.0f;
  if (WS_DOT && g.dot_w) {
    if (pv0) dw0 = *reinterpret_cast<const f4*>(g.dot_w + scol);
    if (pv1) dw1 = *reinterpret_cast<const f4*>(g.dot_w + 64 + scol);
    if (pv2) dw2 = *reinterpret_cast<const f4*>(g.dot_w + 128 + scol);
    if (pv3) dw3 = *reinterpret_cast<const f4*>(g.dot_w + 192 + scol);
    if (g.dot_bias) dot_b = g.dot_bias[0];
  }
  r0a.a = z4; r0a.b = z4; r1a = r0a; r2a = r0a; r3a = r0a; r4a = r0a;
  r0b = r0a; r1b = r0a; r2b = r0a; r3b = r0a; r4b = r0a;
#define WS_FETCH_SET(tile_, S_)                                                   \
  {                                                                               \
    long row_ = WS_TILE(tile_) * WS_TP + srow; if (row_ >= Pn) row_ = Pn - 1;            \
    if (pv0) r0##S_ = view_fetch4(g.A, row_, scol);                               \
    if (pv1) r1##S_ = view_fetch4(g.A, row_, 64 + scol);                          \
    if (pv2) r2##S_ = view_fetch4(g.A, row_, 128 + scol);                         \
    if (pv3) r3##S_ = view_fetch4(g.A, row_, 192 + scol);                         \
    if (K17 && pv4) r4##S_ = view_fetch4(g.A, row_, 256 + scol);                  \
  }
#define WS_FETCH_TILE(tile_) WS_FETCH_SET(tile_, a)
#define WS_PUT_SET(buf_, tile_, S_)                                               \
  {                                                                               \
    const f4 v0 = pv0 ? view_finish4(g.A, r0##S_, scol) : z4;                     \
    const f4 v1 = pv1 ? view_finish4(g.A, r1##S_, 64 + scol) : z4;                \
    const f4 v2 = pv2 ? view_finish4(g.A, r2##S_, 128 + scol) : z4;               \
    const f4 v3 = pv3 ? view_finish4(g.A, r3##S_, 192 + scol) : z4;               \
    const f4 v4 = (K17 && pv4) ? view_finish4(g.A, r4##S_, 256 + scol) : z4;      \
    float mx = fmaxf(fmaxf(fmaxf(ws_absmax4(v0), ws_absmax4(v1)), fmaxf(ws_absmax4(v2), ws_absmax4(v3))), ws_absmax4(v4)); \
    mx = cnr_max16(mx); \
    float sc = 1.0f;                                                              \
    if (mx > 0.0f && mx < 3.0e38f) { int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100; sc = ldexpf(1.0f, 14 - e_); } /* 2^e_ clamp: subnormal rows must not overflow the scale */ \
    unsigned char* dst = smem_b + (buf_) * abuf + srow * ald + scol * 2;          \
    if (pv0) ws_put4(v0, sc, dst, aplane);                                        \
    if (pv1) ws_put4(v1, sc, dst + 128, aplane);                                  \
    if (pv2) ws_put4(v2, sc, dst + 256, aplane);                                  \
    if (pv3) ws_put4(v3, sc, dst + 384, aplane);                                  \
    if (K17 && pv4) ws_put4(v4, sc, dst + 512, aplane);                           \
    float dsum_ = 0.0f;                                                           \
    if (WS_DOT && g.dot_w) {                                                                \
      dsum_ = (ws_dot4(v0, dw0) + ws_dot4(v1, dw1)) + (ws_dot4(v2, dw2) + ws_dot4(v3, dw3)); \
      _Pragma("unroll") for (int d = 8; d >= 1; d >>= 1) dsum_ += __shfl_xor(dsum_, d, 16); \
    }                                                                             \
    if ((tid & 15) == 0) {                                                        \
      reinterpret_cast<float*>(smem_b + (buf_) * abuf + 2 * aplane)[srow] = cnr_pow2_rcp(sc); \
      const long prow_ = WS_TILE(tile_) * WS_TP + srow;                                  \
      if (WS_DOT && g.dot_w && prow_ < Pn) g.dot_out[prow_] = (dsum_ + dot_b) * g.dot_scale; \
      if (g.rs_out && prow_ < Pn) g.rs_out[prow_] = (mx > 0.0f && mx < 3.0e38f) ? sc : (mx == 0.0f ? 0.0f : __builtin_nanf("")); /* 0: all-zero row, NaN: non-finite row (must keep poisoning the weight gradient) */ \
    }                                                                             \
  }
#define WS_PUT_TILE(buf_, tile_) WS_PUT_SET(buf_, tile_, a)
#if WS_GEN_MFMA16
#define WS_MFMA(kb_)                                                                               \
  if (2 * (kb_) < nkb) {                                                                           \
    /* all four fragments of the block first: one exposed LDS round trip per k32 block, not one per row block */ \
    f16x8 a1_[2], a2_[2];                                                                          \
    _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                             \
      a1_[rb] = *reinterpret_cast<const f16x8*>(Ab + rb * 16 * ald + (kb_) * 64);                  \
      a2_[rb] = *reinterpret_cast<const f16x8*>(Ab + rb * 16 * ald + aplane + (kb_) * 64);         \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                             \
      if (2 * (kb_) + 1 == nkb) { if (khi) { a1_[rb] = zero8; a2_[rb] = zero8; } }   /* (uniform test first: the lane select only runs in an odd last block) */ \
      _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                           \
        ws_f32x4 c = acc[rb][cb];                                                                  \
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1_[rb], w2[kb_][cb], c, 0, 0, 0);              \
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2_[rb], w1[kb_][cb], c, 0, 0, 0);              \
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1_[rb], w1[kb_][cb], c, 0, 0, 0);              \
        acc[rb][cb] = c;                                                                           \
      }                                                                                            \
    }                                                                                              \
  }
#else
#define WS_MFMA(kb_)                                                                               \
  if ((kb_) < nkb) {                                                                               \
    const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + (kb_) * 32);                             \
    const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + aplane + (kb_) * 32);                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb_], acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb_], acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb_], acc, 0, 0, 0);                       \
  }
#endif

  // Waves 0..3 (rows 0..15 of a tile) and waves 4..7 (rows 16..31) share the SIMDs pairwise and run half an iteration out of
  // phase: the late group converts + stores its half of tile t+1 (fetched one iteration earlier) and fetches tile t+2 BEFORE
  // its MFMAs of tile t, the early group fetches tile t+1 before and stores it after -- so one wave of each SIMD is in the
  // matrix pipe while the other does prologue math / LDS stores / epilogue.  One barrier per tile.
  const bool late = wave >= 4;
  // MFMAs + epilogue of tile t from LDS buffer buf
  auto compute = [&](const long tc, const int buf) {
    const long t = WS_TILE(tc);
#if WS_GEN_MFMA16
    ws_f32x4 acc[2][2];   // [row block of 16 points][column block of 16 columns]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[rb][cb][j] = 0.0f;
    const unsigned char* Ab = smem_b + buf * abuf + (lane & 15) * ald + (lane >> 4) * 16;
    if (has_w) {
      WS_MFMA(0) WS_MFMA(1) WS_MFMA(2) WS_MFMA(3) WS_MFMA(4) WS_MFMA(5) WS_MFMA(6) WS_MFMA(7)
      if (K17) { WS_MFMA(NKB2 - 1) }
    }
#else
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    const unsigned char* Ab = smem_b + buf * abuf + (lane & 31) * ald + (lane >> 5) * 16;
    if (has_w) {
      WS_MFMA(0) WS_MFMA(1) WS_MFMA(2) WS_MFMA(3) WS_MFMA(4) WS_MFMA(5) WS_MFMA(6) WS_MFMA(7)
      WS_MFMA(8) WS_MFMA(9) WS_MFMA(10) WS_MFMA(11) WS_MFMA(12) WS_MFMA(13) WS_MFMA(14) WS_MFMA(15)
      if (K17) { WS_MFMA(NKB - 1) }
    }
#endif
    // epilogue of this 32 x 32 tile: undo the exact row / column scales, then the fused epilogue on 4 columns per lane.
    // The side inputs of all four row groups are requested first: one memory round trip per tile, and no load has to
    // wait behind the stores of the previous row group.
    if (c0 < ncols_live) {
      const float* rs = reinterpret_cast<const float*>(smem_b + buf * abuf + 2 * aplane);
#if WS_GEN_MFMA16
      {
        const int q4 = lane >> 4, cl = lane & 15;   // result block: rows 4 q4 + r, column cl
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(16 * rb + 4 * q4 + r) * WS_TLD + 16 * cb + cl] = acc[rb][cb][r];
      }
#else
      const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * WS_TLD + cl] = acc[r];
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      constexpr int EG = (EK == EK_SWEEP || EK == EK_VBACK) ? 2 : 4;   // row groups whose side inputs are in flight together (register budget)
#pragma unroll
      for (int i0 = 0; i0 < 4; i0 += EG) {
        EpiRaw4 er[EG];
#pragma unroll
        for (int i = 0; i < EG; ++i) {
          long row = t * WS_TP + (lane >> 3) + 8 * (i0 + i);
          if (row >= Pn) row = Pn - 1;
          if (efast) er[i] = epi_fetch4(g.E, row, ecol);
        }
#pragma unroll
        for (int i = 0; i < EG; ++i) {
          const int rr = (lane >> 3) + 8 * (i0 + i), cc = (lane & 7) * 4;
          const long row = t * WS_TP + rr;
          const float rsc = rs[rr];
          f4 v = *reinterpret_cast<const f4*>(T + rr * WS_TLD + cc);
          v.x *= rsc * wsc.x; v.y *= rsc * wsc.y; v.z *= rsc * wsc.z; v.w *= rsc * wsc.w;
          if (row < Pn) {
            if (efast) epi_finish4(g.E, row, ecol, v, bias4, er[i]);
            else epi_apply4(g.E, row, ecol, v);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  if (!DEEP) {
    WS_FETCH_TILE(t0)
    WS_PUT_TILE(0, t0)
    if (late && t0 + 1 < t1) WS_FETCH_TILE(t0 + 1)
    cnr_lds_barrier();
    for (long t = t0; t < t1; ++t) {
      const int buf = (int)((t - t0) & 1);
      const bool more = t + 1 < t1;
      if (!late) {
        if (more) WS_FETCH_TILE(t + 1)
      } else if (more) {
        WS_PUT_TILE(buf ^ 1, t + 1)
        if (t + 2 < t1) WS_FETCH_TILE(t + 2)
      }
      compute(t, buf);
      if (!late && more) WS_PUT_TILE(buf ^ 1, t + 1)
      cnr_lds_barrier();
    }
  } else {
    // two staging sets: tile t0 + j lives in set a for even j, set b for odd j.  Early waves keep tiles t+1 and t+2 in flight,
    // late waves t+2 and t+3.
    WS_FETCH_SET(t0, a)
    WS_PUT_SET(0, t0, a)
    if (t0 + 1 < t1) WS_FETCH_SET(t0 + 1, b)
    if (late && t0 + 2 < t1) WS_FETCH_SET(t0 + 2, a)
    cnr_lds_barrier();
    for (long t = t0; t < t1; t += 2) {
      {   // even tile t (LDS buffer 0): the next tile waits in set b, set a is free
        const bool more = t + 1 < t1;
        if (!late) {
          if (t + 2 < t1) WS_FETCH_SET(t + 2, a)
        } else if (more) {
          WS_PUT_SET(1, t + 1, b)
          if (t + 3 < t1) WS_FETCH_SET(t + 3, b)
        }
        compute(t, 0);
        if (!late && more) WS_PUT_SET(1, t + 1, b)
        cnr_lds_barrier();
      }
      if (t + 1 < t1) {   // odd tile t + 1 (LDS buffer 1): the next tile waits in set a, set b is free
        const bool more = t + 2 < t1;
        if (!late) {
          if (t + 3 < t1) WS_FETCH_SET(t + 3, b)
        } else if (more) {
          WS_PUT_SET(0, t + 2, a)
          if (t + 4 < t1) WS_FETCH_SET(t + 4, a)
        }
        compute(t + 1, 1);
        if (!late && more) WS_PUT_SET(0, t + 2, a)
        cnr_lds_barrier();
      }
    }
  }
#undef WS_FETCH_TILE
#undef WS_PUT_TILE
#undef WS_FETCH_SET
#undef WS_PUT_SET
#undef WS_MFMA
#undef WS_TILE
}

// walk-direction parity of consecutive layer launches: process-wide, atomic (host threads may launch concurrently); it only picks the
// order in which a launch visits its tiles, never a value
inline int ws_next_rev() { static std::atomic<unsigned> parity{0}; return (int)(parity.fetch_add(1u, std::memory_order_relaxed) & 1u); }

// ================================================================================================
// "Stream" form of the same kernel for the launches that need none of its special cases: K in (240, 256], 256 live output columns
// whose epilogue takes the 16-byte path in every lane, no tail fill, no split point, a point count that is a multiple of the tile
// height and known on the host (ws_stream_ok).
// Same tiles, same arithmetic, same order of operations per output element -- what changes is the control flow around the memory
// operations: one loop per wave group (early / late), prefetches issued unconditionally (tile index clamped to the range), first
// iterations peeled.  Every path through a loop then issues the same memory operations in the same order, and the compiler's
// s_waitcnt bookkeeping stays exact: where a staging register set is consumed it waits for THAT set (vmcnt(8..15)) instead of falling back
// to vmcnt(0), which made every wave wait once per tile for its just-issued epilogue stores and for the deeper prefetch
// (tools/probes/ws_depth_probe.hip -> ws_depth3_probe.hip: 0.313 -> 0.280 ms on a plain layer).
// ================================================================================================
// Branch-free forms of epi_fetch4 / epi_finish4 for an epilogue without split point and tail fill (the stream kernel's promise): a branch
// between two memory operations costs the compiler its count of what is still in flight.
// The epilogue's side inputs (pre-activations, ReLU masks, the cotangent a value-backward launch adds to) are read once per launch: as NON-TEMPORAL loads they do not
// push the lines the launch re-reads (its own input tile, the weights) or hands to the next launch (its output rows) out of L2 / Infinity Cache: +0.2-0.3 % of the step,
// same bits (profiles/r06_ab_nt_side_inputs.txt).  The STAGED tile must stay a plain load: non-temporal there costs 2 % (the fused launches read it twice).
// ... and the second-order cotangent a sweep launch writes on z (picked up by the value-backward chain many launches later) as a non-temporal store: +0.2 %
#ifndef WS_NT_SWEEP
#define WS_NT_SWEEP 1
#endif
#ifndef WS_NT_SIDE
#define WS_NT_SIDE 1
#endif
#if WS_NT_SIDE
#define WS_NTLOAD(p_) __builtin_nontemporal_load(reinterpret_cast<const f4*>(p_))
#else
#define WS_NTLOAD(p_) (*reinterpret_cast<const f4*>(p_))
#endif
template <int EK>
__device__ __forceinline__ EpiRaw4 epi_fetch4_plain(const Epi& e, long row, int col) {
  EpiRaw4 r;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  r.a = zero; r.b = zero;
  if constexpr (EK == EK_SWEEP) {
    r.a = WS_NTLOAD(e.z + row * e.ldz + col);
    r.b = WS_NTLOAD(e.v + row * e.ldv + col);   // ldv == 0: the broadcast row
  } else if constexpr (EK == EK_VBACK) {
    r.a = WS_NTLOAD(e.z + row * e.ldz + col);
    r.b = WS_NTLOAD(e.o1 + row * e.ld1 + col);
  } else if constexpr (EK == EK_RELU_MASK) {
    r.a = WS_NTLOAD(e.aux + row * e.ldaux + col);
  }
  return r;
}
template <int EK>
__device__ __forceinline__ void epi_finish4_plain(const Epi& e, long row, int col, const f4& acc, const f4& b, const EpiRaw4& raw) {
  if constexpr (EK == EK_SWEEP) {
    const f4 zz = raw.a, vv = raw.b;
    f4 o1, o2;
    o1.x = softplus100_d2(zz.x) * (vv.x * e.vscale) * acc.x; o2.x = softplus100_d1(zz.x) * acc.x;
    o1.y = softplus100_d2(zz.y) * (vv.y * e.vscale) * acc.y; o2.y = softplus100_d1(zz.y) * acc.y;
    o1.z = softplus100_d2(zz.z) * (vv.z * e.vscale) * acc.z; o2.z = softplus100_d1(zz.z) * acc.z;
    o1.w = softplus100_d2(zz.w) * (vv.w * e.vscale) * acc.w; o2.w = softplus100_d1(zz.w) * acc.w;
#if WS_NT_SWEEP
    __builtin_nontemporal_store(o1, reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col));   // (the second-order cotangent on z: picked up by the value-backward chain many launches later)
#else
    *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o1;
#endif
    *reinterpret_cast<f4*>(e.o2 + row * e.ld2 + col) = o2;
  } else if constexpr (EK == EK_VBACK) {
    const f4 zz = raw.a;
    f4 o = raw.b;
    o.x = softplus100_d1(zz.x) * (acc.x * e.scale) + o.x;
    o.y = softplus100_d1(zz.y) * (acc.y * e.scale) + o.y;
    o.z = softplus100_d1(zz.z) * (acc.z * e.scale) + o.z;
    o.w = softplus100_d1(zz.w) * (acc.w * e.scale) + o.w;
    *reinterpret_cast<f4*>(e.o1 + row * e.ld1 + col) = o;
  } else {
    epi_finish4(e, row, col, acc, b, raw);
  }
}

template <int EK, bool GEN>
__device__ __forceinline__ EpiRaw4 epi_fetch4_sel(const Epi& e, long row, int col) {
  if constexpr (GEN) return epi_fetch4(e, row, col); else return epi_fetch4_plain<EK>(e, row, col);
}
template <int EK, bool GEN>
__device__ __forceinline__ void epi_finish4_sel(const Epi& e, long row, int col, const f4& acc, const f4& b, const EpiRaw4& raw) {
  if constexpr (GEN) epi_finish4(e, row, col, acc, b, raw); else epi_finish4_plain<EK>(e, row, col, acc, b, raw);
}

// GEN: the epilogue keeps its tail fill / split point (skip-connection layers); it then runs the general 16-byte epilogue code per lane.
template <int VK, int EK, bool GEN = false>
__global__ __launch_bounds__(WS_THREADS, 1) void layer_gemm_ws_stream_kernel(const LayerGemm g_in, int tiles_per_wg, int rev) {
  constexpr int NKB = 16;
  LayerGemm g = g_in;
  g.A.kind = VK; g.E.kind = EK;
  if (!GEN) { g.E.tail_src = nullptr; g.E.tail_n = 0; g.E.split = 1 << 30; }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long Pn = g.P;                                  // a multiple of the tile height, known on the host (ws_stream_ok)
  const long ntiles = Pn / WS_TP;
  const long t0 = (long)blockIdx.x * tiles_per_wg;
  if (t0 >= ntiles) return;
  long t1 = t0 + tiles_per_wg;
  if (t1 > ntiles) t1 = ntiles;
  const long tflip = t0 + t1 - 1, tlast = t1 - 1;
#define WSS_TILE(t_) (rev ? tflip - (t_) : (t_))
  constexpr int kpad = NKB * 16;
  constexpr int ald = kpad * 2 + 16;
  constexpr int aplane = WS_TP * ald;
  constexpr int abuf = 2 * aplane + 128;
  float* T = reinterpret_cast<float*>(smem_b + 2 * abuf) + wave * (32 * WS_TLD);
  const int c0 = g.col0 + wave * 32;

#if WS_MFMA16
  // (round 6) the stream form on v_mfma_f32_16x16x32_f16 -- 12 % cheaper per FLOP than the 32 x 32 x 16 shape and a 15 % higher clock under the board's power
  // limit (profiles/r06_mfma_shapes.txt): B fragments of this wave's 2 x 16 output columns, lane (n = lane & 15, kg = lane >> 4) holds W[c0 + 16 cb + n][32 kb + 8 kg ..]
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
  const f4 wsc = *reinterpret_cast<const f4*>(g.wscale + c0 + (lane & 7) * 4);
  const int ecol = c0 + (lane & 7) * 4;
  const f4 bias4 = epi_bias4(g.E, ecol);
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  // second staging set (tiles fetched two ahead) where the registers allow it without spilling: the softplus prologue needs the room
  constexpr bool DEEP = VK == VK_DIRECT && (EK == EK_STORE || EK == EK_RELU || EK == EK_SDF_TOP);
  Raw4 r0a, r1a, r2a, r3a, r0b, r1b, r2b, r3b;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  f4 dw0 = z4, dw1 = z4, dw2 = z4, dw3 = z4;      // row-dot weights (LayerGemm::dot_w): only the top SDF layer's instantiation carries them
  float dot_b = 0.0f;
  if (EK == EK_SDF_TOP && g.dot_w) {
    dw0 = *reinterpret_cast<const f4*>(g.dot_w + scol);
    dw1 = *reinterpret_cast<const f4*>(g.dot_w + 64 + scol);
    dw2 = *reinterpret_cast<const f4*>(g.dot_w + 128 + scol);
    dw3 = *reinterpret_cast<const f4*>(g.dot_w + 192 + scol);
    if (g.dot_bias) dot_b = g.dot_bias[0];
  }
  r0a.a = z4; r0a.b = z4; r1a = r0a; r2a = r0a; r3a = r0a;
  r0b = r0a; r1b = r0a; r2b = r0a; r3b = r0a;
  // (tile_ may run past the range: clamped, the redundant fetch / LDS tile of the last iterations is never used)
#define WSS_FETCH(tile_, S_)                                                      \
  {                                                                               \
    const long tq_ = (tile_) < tlast ? (tile_) : tlast;                           \
    const long row_ = WSS_TILE(tq_) * WS_TP + srow;                               \
    r0##S_ = view_fetch4(g.A, row_, scol);                                        \
    r1##S_ = view_fetch4(g.A, row_, 64 + scol);                                   \
    r2##S_ = view_fetch4(g.A, row_, 128 + scol);                                  \
    r3##S_ = view_fetch4(g.A, row_, 192 + scol);                                  \
    __builtin_amdgcn_sched_barrier(0);                                            \
  }
#define WSS_PUT(buf_, tile_, S_)                                                  \
  {                                                                               \
    __builtin_amdgcn_s_setprio(2);   /* the conversion + LDS stores of the next tile gate every wave's next barrier (-0.05 ms per step) */ \
    const f4 v0 = view_finish4(g.A, r0##S_, scol);                                \
    const f4 v1 = view_finish4(g.A, r1##S_, 64 + scol);                           \
    const f4 v2 = view_finish4(g.A, r2##S_, 128 + scol);                          \
    const f4 v3 = view_finish4(g.A, r3##S_, 192 + scol);                          \
    float mx = fmaxf(fmaxf(ws_absmax4(v0), ws_absmax4(v1)), fmaxf(ws_absmax4(v2), ws_absmax4(v3))); \
    mx = cnr_max16(mx); \
    float sc = 1.0f;                                                              \
    if (mx > 0.0f && mx < 3.0e38f) { int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100; sc = ldexpf(1.0f, 14 - e_); } \
    unsigned char* dst = smem_b + (buf_) * abuf + srow * ald + scol * 2;          \
    ws_put4(v0, sc, dst, aplane);                                                 \
    ws_put4(v1, sc, dst + 128, aplane);                                           \
    ws_put4(v2, sc, dst + 256, aplane);                                           \
    ws_put4(v3, sc, dst + 384, aplane);                                           \
    float dsum_ = 0.0f;                                                           \
    if (EK == EK_SDF_TOP && g.dot_w) {                                            \
      dsum_ = (ws_dot4(v0, dw0) + ws_dot4(v1, dw1)) + (ws_dot4(v2, dw2) + ws_dot4(v3, dw3)); \
      _Pragma("unroll") for (int d = 8; d >= 1; d >>= 1) dsum_ += __shfl_xor(dsum_, d, 16); \
    }                                                                             \
    if ((tid & 15) == 0) {                                                        \
      reinterpret_cast<float*>(smem_b + (buf_) * abuf + 2 * aplane)[srow] = cnr_pow2_rcp(sc); \
      const long prow_ = WSS_TILE((tile_) < tlast ? (tile_) : tlast) * WS_TP + srow; \
      if (EK == EK_SDF_TOP && g.dot_w) g.dot_out[prow_] = (dsum_ + dot_b) * g.dot_scale; \
      if (g.rs_out) g.rs_out[prow_] = (mx > 0.0f && mx < 3.0e38f) ? sc : (mx == 0.0f ? 0.0f : __builtin_nanf("")); \
    }                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                            \
    __builtin_amdgcn_s_setprio(0);                                                \
  }
  const bool late = wave >= 4;
  constexpr bool EPRE = EK == EK_VBACK || EK == EK_RELU_MASK || EK == EK_SWEEP;
  EpiRaw4 ern[4];
  if (EPRE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ern[i] = epi_fetch4_sel<EK, GEN>(g.E, WSS_TILE(t0) * WS_TP + (lane >> 3) + 8 * i, ecol);
  }
  auto compute = [&](const long tc, const int buf) {
    const long t = WSS_TILE(tc < tlast ? tc : tlast);
#if WS_MFMA16
    ws_f32x4 acc[2][2];   // [row block of 16 points][column block of 16 columns]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[rb][cb][j] = 0.0f;
    const unsigned char* Ab = smem_b + buf * abuf + (lane & 15) * ald + (lane >> 4) * 16;   // A fragment: tile[16 rb + (lane & 15)][32 kb + 8 (lane >> 4) ..]
#pragma unroll
    for (int kb = 0; kb < NKB2; ++kb) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + rb * 16 * ald + kb * 64);
        const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + rb * 16 * ald + aplane + kb * 64);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          ws_f32x4 c = acc[rb][cb];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w2[kb][cb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, w1[kb][cb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w1[kb][cb], c, 0, 0, 0);
          acc[rb][cb] = c;
        }
      }
    }
    // (scheduling fences between the phases: the blocks are branch-free now, and an unconstrained scheduler hoists the next phase's loads
    // across the MFMA block until the register file spills)
    __builtin_amdgcn_sched_barrier(0);
    const float* rs = reinterpret_cast<const float*>(smem_b + buf * abuf + 2 * aplane);
    {
      const int q4 = lane >> 4, cl = lane & 15;   // result block: rows 4 q4 + r, column cl
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(16 * rb + 4 * q4 + r) * WS_TLD + 16 * cb + cl] = acc[rb][cb][r];
    }
#else
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    const unsigned char* Ab = smem_b + buf * abuf + (lane & 31) * ald + (lane >> 5) * 16;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32);
      const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + aplane + kb * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0);
    }
    // (scheduling fences between the phases: the blocks are branch-free now, and an unconstrained scheduler hoists the next phase's loads
    // across the MFMA block until the register file spills)
    __builtin_amdgcn_sched_barrier(0);
    const float* rs = reinterpret_cast<const float*>(smem_b + buf * abuf + 2 * aplane);
    const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * WS_TLD + cl] = acc[r];
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (EPRE) {
      // side inputs requested one tile ahead (ern): when they are consumed here they are the OLDEST loads in flight, so the wait
      // leaves the prefetched activation tile and the stores of the previous tiles alone (a load consumed right after its issue is the
      // youngest one, and the in-order counter then drains everything before it: twice per tile in the general kernel)
      const long tn = WSS_TILE(tc + 1 < tlast ? tc + 1 : tlast);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = (lane >> 3) + 8 * i, cc = (lane & 7) * 4;
        const long row = t * WS_TP + rr;
        const float rsc = rs[rr];
        f4 v = *reinterpret_cast<const f4*>(T + rr * WS_TLD + cc);
        v.x *= rsc * wsc.x; v.y *= rsc * wsc.y; v.z *= rsc * wsc.z; v.w *= rsc * wsc.w;
        epi_finish4_sel<EK, GEN>(g.E, row, ecol, v, bias4, ern[i]);
        ern[i] = epi_fetch4_sel<EK, GEN>(g.E, tn * WS_TP + rr, ecol);
      }
    } else {
    constexpr int EG = (EK == EK_SWEEP || EK == EK_VBACK) ? 2 : 4;
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += EG) {
      EpiRaw4 er[EG];
#pragma unroll
      for (int i = 0; i < EG; ++i) {
        er[i] = epi_fetch4_sel<EK, GEN>(g.E, t * WS_TP + (lane >> 3) + 8 * (i0 + i), ecol);
      }
#pragma unroll
      for (int i = 0; i < EG; ++i) {
        const int rr = (lane >> 3) + 8 * (i0 + i), cc = (lane & 7) * 4;
        const long row = t * WS_TP + rr;
        const float rsc = rs[rr];
        f4 v = *reinterpret_cast<const f4*>(T + rr * WS_TLD + cc);
        v.x *= rsc * wsc.x; v.y *= rsc * wsc.y; v.z *= rsc * wsc.z; v.w *= rsc * wsc.w;
        epi_finish4_sel<EK, GEN>(g.E, row, ecol, v, bias4, er[i]);   // (no row test: every tile is full, so no branch sits between the memory operations)
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // LDS buffer of relative tile j is j & 1.  All waves execute one barrier before the loops and one per tile.
  if (!DEEP) {
    WSS_FETCH(t0, a)
    WSS_PUT(0, t0, a)
    if (!late) {
      cnr_lds_barrier();
      for (long t = t0; t < t1; ++t) {
        const int buf = (int)((t - t0) & 1);
        WSS_FETCH(t + 1, a)
        compute(t, buf);
        WSS_PUT(buf ^ 1, t + 1, a)
        cnr_lds_barrier();
      }
    } else {
      WSS_FETCH(t0 + 1, a)
      cnr_lds_barrier();
#define WSS_LATE1(t_)                                                             \
      {                                                                           \
        const int buf = (int)(((t_) - t0) & 1);                                   \
        WSS_PUT(buf ^ 1, (t_) + 1, a)                                             \
        WSS_FETCH((t_) + 2, a)                                                    \
        compute((t_), buf);                                                       \
        cnr_lds_barrier();                                                        \
      }
      WSS_LATE1(t0)   // peeled: the loop header then only sees the steady state
      for (long t = t0 + 1; t < t1; ++t) WSS_LATE1(t)
#undef WSS_LATE1
    }
  } else {
    // two staging sets: relative tile j lives in set a for even j, set b for odd j; early waves keep tiles t + 1 and t + 2 in flight,
    // late waves t + 2 and t + 3.  Two tiles per trip without a branch in between: the trailing half-trip of an odd range repeats the
    // last tile (clamped index; these epilogues are plain stores, so writing the same values twice is harmless).
    WSS_FETCH(t0, a)
    WSS_PUT(0, t0, a)
    WSS_FETCH(t0 + 1, b)
    if (!late) {
      cnr_lds_barrier();
#define WSS_EARLY2(t_)                                                            \
      {                                                                           \
        WSS_FETCH((t_) + 2, a)                                                    \
        compute((t_), 0);                                                         \
        WSS_PUT(1, (t_) + 1, b)                                                   \
        cnr_lds_barrier();                                                        \
        WSS_FETCH((t_) + 3, b)                                                    \
        compute((t_) + 1, 1);                                                     \
        WSS_PUT(0, (t_) + 2, a)                                                   \
        cnr_lds_barrier();                                                        \
      }
      WSS_EARLY2(t0)
      for (long t = t0 + 2; t < t1; t += 2) WSS_EARLY2(t)
#undef WSS_EARLY2
    } else {
      WSS_FETCH(t0 + 2, a)
      cnr_lds_barrier();
#define WSS_LATE2(t_)                                                             \
      {                                                                           \
        WSS_PUT(1, (t_) + 1, b)                                                   \
        WSS_FETCH((t_) + 3, b)                                                    \
        compute((t_), 0);                                                         \
        cnr_lds_barrier();                                                        \
        WSS_PUT(0, (t_) + 2, a)                                                   \
        WSS_FETCH((t_) + 4, a)                                                    \
        compute((t_) + 1, 1);                                                     \
        cnr_lds_barrier();                                                        \
      }
      WSS_LATE2(t0)
      for (long t = t0 + 2; t < t1; t += 2) WSS_LATE2(t)
#undef WSS_LATE2
    }
  }
#undef WSS_FETCH
#undef WSS_PUT
#undef WSS_TILE
}

// ---- MFMA fragments out of ROW-MAJOR [point][column] f16 planes by the LDS transpose read (gfx950 ds_read_b64_tr_b16): a 16-lane group reads a
// [4 points][16 columns] block, 8 contiguous bytes per lane (lane i: block row i >> 2, columns 4 (i & 3) ..), and lane i receives column i of it.
// Two reads make one fragment (8 k positions of the lane's row / column).  The four rows of a block are taken 4 POINTS APART: with a row stride of
// 4 banks mod 64 (528 B, 272 B, 144 B) they sit 16 banks apart and the 2 x 4 x 4 eight-byte pieces of a 32-lane half cover the 64 banks exactly
// once.  So position q = 4 h + r of k group kg in k16 block kb holds point 16 kb + 4 r + 2 kg + h (ws_kslot is the inverse) -- for both operands of
// a product over points, which is all the MFMA needs.  `src`: the lane's piece for h = 0 (point row kb * 16 + 2 kg + 4 ((lane & 15) >> 2), columns
// base + (lane & 16) + 4 (lane & 3)); the piece for h = 1 lies one point row (ld bytes) below.
typedef short ws_s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ws_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 ws_tr8(const unsigned char* src, int ld) {
  typedef __attribute__((address_space(3))) ws_s16x4* lds_s16x4;
  const ws_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(src));
  const ws_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(src + ld));
  return __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int ws_kslot(int pt) {   // k position (within a 32-point tile) at which the fragments above hold point pt
  return (pt & 16) | ((pt & 2) << 2) | ((pt & 1) << 2) | ((pt >> 2) & 3);
}

// host-side test for the stream form: every wave has weights and a live 32-column epilogue block, every lane the 16-byte epilogue path
inline bool ws_stream_ok(const LayerGemm& g, int wrows) {
  const bool off = debug_flags().ws_nostream;   // debugging aid: the general kernel for every launch
  if (off || g.K <= 240 || g.K > 256 || g.P_dev != nullptr || (g.P % WS_TP) != 0) return false;
  if (g.dot_w != nullptr && g.E.kind != EK_SDF_TOP) return false;   // (the row dot lives in that instantiation of the stream form only)
  const int live = g.E.n_out + (g.E.tail_src ? g.E.tail_n : 0);
  if (g.col0 + 256 > wrows || g.col0 + 256 > live) return false;
  for (int c = g.col0; c < g.col0 + 256; c += 4) if (!epi_fast4(g.E, c)) return false;
  return true;
}
inline bool ws_stream_plain(const LayerGemm& g) { return g.E.tail_src == nullptr && g.E.split == (1 << 30); }

template <int VK, int EK, bool GEN>
static void launch_ws_stream(const LayerGemm& g, cnr_stream s) {
  constexpr int abuf = 2 * WS_TP * (16 * 32 + 16) + 128;
  const size_t lds = (size_t)2 * abuf + (size_t)8 * 32 * WS_TLD * sizeof(float);
  const long ntiles = (g.P + WS_TP - 1) / WS_TP;
  if (ntiles == 0) return;
  const int ws_wgs = debug_flags().ws_wgs;
  long tpw = (ntiles + ws_wgs - 1) / ws_wgs;
  if (tpw < 1) tpw = 1;
  const unsigned grid = (unsigned)((ntiles + tpw - 1) / tpw);
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_gemm_ws_stream_kernel<VK, EK, GEN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int ws_serp = debug_flags().ws_serp;
  const int rev = ws_serp ? ws_next_rev() : 0;
  TimingScope ts_("layer_gemm_ws", 0, 100 + (g.N + 31) / 32, g.P, g.N, g.K, 1, s, layer_gemm_bytes(g));
  hipLaunchKernelGGL((layer_gemm_ws_stream_kernel<VK, EK, GEN>), dim3(grid), dim3(WS_THREADS), lds, s, g, (int)tpw, rev);
}

template <int VK, int EK, bool PLAIN, bool K17, bool FULLK>
static void launch_ws_tk(const LayerGemm& g, int wrows, cnr_stream s) {
  const int nkb = (g.K + 15) / 16;
  const int abuf = 2 * WS_TP * (nkb * 32 + 16) + 128;
  const size_t lds = (size_t)2 * abuf + (size_t)8 * 32 * WS_TLD * sizeof(float);
  const long ntiles = (g.P + WS_TP - 1) / WS_TP;
  if (ntiles == 0) return;
  const int ws_wgs = debug_flags().ws_wgs;       // tuning knobs (defaults measured on MI355X: 256 / 1)
  const int ws_mintpw = debug_flags().ws_mintpw;
  long tpw = (ntiles + ws_wgs - 1) / ws_wgs;   // one workgroup per CU: the weights are loaded once per CU (measured best of 256 / 512 / 768 / 1024)
  if (tpw < ws_mintpw) tpw = ws_mintpw;
  const unsigned grid = (unsigned)((ntiles + tpw - 1) / tpw);
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_gemm_ws_kernel<VK, EK, PLAIN, K17, FULLK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int ws_serp = debug_flags().ws_serp;   // alternate walk direction: +0.35 % (A/B, the last rows of a producer are still in L2)
  const int rev = ws_serp ? ws_next_rev() : 0;
  TimingScope ts_("layer_gemm_ws", 0, 100 + (g.N + 31) / 32, g.P, g.N, g.K, 1, s, layer_gemm_bytes(g));
  hipLaunchKernelGGL((layer_gemm_ws_kernel<VK, EK, PLAIN, K17, FULLK>), dim3(grid), dim3(WS_THREADS), lds, s, g, (int)tpw, wrows, rev);
}

template <int VK, int EK, bool PLAIN, bool K17 = false>
static void launch_ws_t(const LayerGemm& g, int wrows, cnr_stream s) {
  const int nkb = (g.K + 15) / 16;
  if constexpr (!K17 && VK >= 0 && EK >= 0) {
    if (ws_stream_ok(g, wrows) && ws_stream_plain(g) == PLAIN) { launch_ws_stream<VK, EK, !PLAIN>(g, s); return; }
  }
  if (nkb == (K17 ? 17 : 16)) launch_ws_tk<VK, EK, PLAIN, K17, true>(g, wrows, s);
  else launch_ws_tk<VK, EK, PLAIN, K17, false>(g, wrows, s);
}

}  // namespace cnr
