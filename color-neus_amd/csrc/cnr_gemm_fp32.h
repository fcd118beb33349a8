// FP32-MFMA layer GEMM: kernel template and launcher (instantiated in cnr_gemm.hip for 1..4 column tiles, cnr_gemm_wide.hip for 5..8).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"

namespace cnr {

// ================================================================================================
// layer GEMM
// ================================================================================================
constexpr int LG_BM = 128;     // points per workgroup (4 waves x 32 rows)
constexpr int LG_BK = 16;      // K slab
constexpr int LG_LD = 20;      // LDS row stride in floats: 20 = 4*5 -> ds_read_b128 of 16 rows hits 64 distinct banks
constexpr int LG_TLD = 36;     // row stride of the epilogue transpose tile (16-byte aligned rows -> ds_read_b128)

// compile-time recursion over the N tiles: accumulator indices stay static (a runtime-indexed ext-vector array would be
// placed in scratch memory)
template <int NT, int I>
__device__ __forceinline__ void lg_epilogue_tiles(const f32x16 (&acc)[NT], const Epi& e, float* T, long wave_row0, long P, int lane,
                                                  int ncols_live, int col0) {
  if constexpr (I < NT) {
    if (col0 + I * 32 < ncols_live) {
      const int hi = lane >> 5, cl = lane & 31;
      const int er = lane >> 3, ec4 = (lane & 7) * 4;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * LG_TLD + cl] = acc[I][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = er + 8 * i;
        const long row = wave_row0 + rr;
        const f4 v = *reinterpret_cast<const f4*>(T + rr * LG_TLD + ec4);
        if (row < P) epi_apply4(e, row, col0 + I * 32 + ec4, v);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    lg_epilogue_tiles<NT, I + 1>(acc, e, T, wave_row0, P, lane, ncols_live, col0);
  }
}

// VK / EK >= 0 pin the view / epilogue kind at compile time (used for the narrow launches of the render plan)
template <int NT, int VK = -1, int EK = -1>
__global__ __launch_bounds__(256, 2) void layer_gemm_kernel(const LayerGemm g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                            // [2][128][LG_LD]
  float* Bs = smem + 2 * LG_BM * LG_LD;        // [2][NT*32][LG_LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * LG_BM;
  const long Pn = g.P_dev ? (long)*g.P_dev : g.P;   // compacted inputs: tiles past the device-side count have nothing to do
  if (row0 >= Pn) return;
  const int nslab = (g.K + LG_BK - 1) / LG_BK;
  constexpr int NB = (NT * 32 * 4 + 255) / 256;   // float4 of W per thread per slab
  constexpr bool NB_EXACT = (NT * 32 * 4) % 256 == 0;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;

  // staging registers (native vectors: stay in VGPRs)
  f4 ra0, ra1, rb0a, rb1a;   // A: raw a / b operands of the two float4 this thread stages
  f4 rw[NB];
  const int ar0 = tid >> 2, ar1 = (tid + 256) >> 2, ac4 = (tid & 3) * 4;
  long arow0 = row0 + ar0, arow1 = row0 + ar1;
  if (arow0 >= Pn) arow0 = Pn - 1;             // clamp: rows beyond P are computed on valid data and dropped in the epilogue
  if (arow1 >= Pn) arow1 = Pn - 1;
  View A = g.A;
  if (VK >= 0) A.kind = VK;
  const bool has_b = A.kind == VK_SIGMUL || A.kind == VK_SIGMUL_ROW || A.kind == VK_RELUGATE;
  const float* a0p = A.a + arow0 * A.lda + ac4;
  const float* a1p = A.a + arow1 * A.lda + ac4;
  const float* b0p = (A.kind == VK_SIGMUL || A.kind == VK_RELUGATE) ? A.b + arow0 * A.ldb + ac4 : (A.kind == VK_SIGMUL_ROW ? A.b + ac4 : a0p);
  const float* b1p = (A.kind == VK_SIGMUL || A.kind == VK_RELUGATE) ? A.b + arow1 * A.ldb + ac4 : (A.kind == VK_SIGMUL_ROW ? A.b + ac4 : a1p);
  const float* wp = g.W + (long)(g.col0 + (tid >> 2)) * g.ldw + ac4;

#define LG_LOAD_SLAB(s_)                                                                     \
  {                                                                                          \
    const int ko_ = (s_) * LG_BK;                                                            \
    ra0 = *reinterpret_cast<const f4*>(a0p + ko_);                                           \
    ra1 = *reinterpret_cast<const f4*>(a1p + ko_);                                           \
    if (has_b) {                                                                             \
      rb0a = *reinterpret_cast<const f4*>(b0p + ko_);                                        \
      rb1a = *reinterpret_cast<const f4*>(b1p + ko_);                                        \
    }                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                         \
      if (NB_EXACT || tid + i * 256 < NT * 32 * 4)                                           \
        rw[i] = *reinterpret_cast<const f4*>(wp + (long)i * 64 * g.ldw + ko_);               \
    }                                                                                        \
  }
#define LG_STORE_SLAB(buf_, s_)                                                              \
  {                                                                                          \
    const int kc_ = (s_) * LG_BK + ac4;                                                      \
    Raw4 q0_, q1_;                                                                           \
    q0_.a = ra0; q0_.b = has_b ? rb0a : ra0;                                                 \
    q1_.a = ra1; q1_.b = has_b ? rb1a : ra1;                                                 \
    *reinterpret_cast<f4*>(As + ((buf_) * LG_BM + ar0) * LG_LD + ac4) = view_finish4(A, q0_, kc_); \
    *reinterpret_cast<f4*>(As + ((buf_) * LG_BM + ar1) * LG_LD + ac4) = view_finish4(A, q1_, kc_); \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                         \
      if (NB_EXACT || tid + i * 256 < NT * 32 * 4)                                           \
        *reinterpret_cast<f4*>(Bs + ((buf_) * NT * 32 + (tid >> 2) + i * 64) * LG_LD + ac4) = rw[i]; \
    }                                                                                        \
  }

  LG_LOAD_SLAB(0)
  LG_STORE_SLAB(0, 0)
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) LG_LOAD_SLAB(s + 1)
    const float* Ab = As + (buf * LG_BM + wave * 32 + (lane & 31)) * LG_LD + (lane >> 5) * 4;
    const float* Bb = Bs + (buf * NT * 32 + (lane & 31)) * LG_LD + (lane >> 5) * 4;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const f4 a = *reinterpret_cast<const f4*>(Ab + kb * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4*>(Bb + nt * 32 * LG_LD + kb * 8);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
      }
    }
    if (s + 1 < nslab) LG_STORE_SLAB(buf ^ 1, s + 1)
    __syncthreads();
  }
#undef LG_LOAD_SLAB
#undef LG_STORE_SLAB

  // ---- epilogue: transpose each 32x32 accumulator tile through a wave-private LDS tile, then 4 columns per lane
  float* T = smem + wave * (32 * LG_TLD);
  Epi e = g.E;
  if (EK >= 0) e.kind = EK;
  const int ncols_live = e.n_out + (e.tail_src ? e.tail_n : 0);
  lg_epilogue_tiles<NT, 0>(acc, e, T, row0 + wave * 32, Pn, lane, ncols_live, g.col0);
}

template <int NT, int VK = -1, int EK = -1>
void launch_layer_gemm(const LayerGemm& g, cnr_stream s) {
  const size_t lds = (size_t)(2 * LG_BM * LG_LD + 2 * NT * 32 * LG_LD) * sizeof(float);
  const unsigned grid = (unsigned)((g.P + LG_BM - 1) / LG_BM);
  if (grid == 0) return;
  static DeviceOnce attr_once;   // the opt-in is per device
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_gemm_kernel<NT, VK, EK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  TimingScope ts_("layer_gemm", 0, NT, g.P, g.N, g.K, 1, s, layer_gemm_bytes(g));
  hipLaunchKernelGGL((layer_gemm_kernel<NT, VK, EK>), dim3(grid), dim3(256), lds, s, g);
}


}  // namespace cnr
