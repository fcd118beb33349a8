// FP32-MFMA GEMM kernels for gfx950 (MI355X / CDNA4): the fully-connected layers of the SDF / colour / relight stacks
// (forward, input-gradient chain, second-order sweep, backward) and their weight gradients.
//
//   layer_gemm_kernel<NT>   C[P x N] = epilogue(A[P x K] * W[N x K]^T): 128-point tile x full N per workgroup (4 waves x
//                           32 rows, v_mfma_f32_32x32x2_f32, NT*16 accumulator registers per lane), K streamed in 16-wide
//                           slabs through double-buffered LDS (row stride 20 floats -> conflict-free ds_read_b128), operand
//                           prologue fused into the HBM->LDS staging (cnr_views.h), epilogue fused on the accumulators.
//   dw_gemm_kernel<...>     dW[N x K] = sum_pts X[pt][n] * Y[pt][k]: 8 waves, 256x256 output tile held in registers,
//                           points streamed 16 at a time, per-chunk partial results (deterministic reduction afterwards).
#include <hip/hip_runtime.h>

#include "cnr_backend.h"
#include "cnr_hip_util.h"

namespace cnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ================================================================================================
// layer GEMM
// ================================================================================================
constexpr int LG_BM = 128;     // points per workgroup (4 waves x 32 rows)
constexpr int LG_BK = 16;      // K slab
constexpr int LG_LD = 20;      // LDS row stride in floats: 20 = 4*5 -> ds_read_b128 of 16 rows hits 64 distinct banks

template <int KIND, int NT>
__device__ __forceinline__ void lg_epilogue(const Epi& e, const f32x16 (&acc)[NT], long row_base, int lane, long P);

template <int NT>
__global__ __launch_bounds__(256, (NT <= 8 ? 2 : 1)) void layer_gemm_kernel(const LayerGemm g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                            // [2][128][LG_LD]
  float* Bs = smem + 2 * LG_BM * LG_LD;        // [2][NT*32][LG_LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * LG_BM;
  const int nslab = (g.K + LG_BK - 1) / LG_BK;
  constexpr int NB = (NT * 32 * 4 + 255) / 256;   // float4 of W per thread per slab

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;

  f4 ra[2];
  f4 rb[NB];
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_slab = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int idx = tid + i * 256;
      int r = idx >> 2, c4 = idx & 3;
      long row = row0 + r;
      ra[i] = row < g.P ? view_eval4(g.A, row, s * LG_BK + c4 * 4) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int idx = tid + i * 256;
      int r = idx >> 2, c4 = idx & 3;
      if (NT * 32 * 4 % 256 == 0 || idx < NT * 32 * 4)
        rb[i] = *reinterpret_cast<const f4*>(g.W + (long)r * g.ldw + s * LG_BK + c4 * 4);
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int idx = tid + i * 256;
      int r = idx >> 2, c4 = idx & 3;
      *reinterpret_cast<f4*>(As + (buf * LG_BM + r) * LG_LD + c4 * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int idx = tid + i * 256;
      int r = idx >> 2, c4 = idx & 3;
      if (NT * 32 * 4 % 256 == 0 || idx < NT * 32 * 4)
        *reinterpret_cast<f4*>(Bs + (buf * NT * 32 + r) * LG_LD + c4 * 4) = rb[i];
    }
  };

  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) load_slab(s + 1);
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
    if (s + 1 < nslab) store_slab(buf ^ 1);
    __syncthreads();
  }

  const long row_base = row0 + wave * 32 + 4 * (lane >> 5);
  switch (g.E.kind) {
    case EK_STORE: lg_epilogue<EK_STORE, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_SPLIT: lg_epilogue<EK_SPLIT, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_SDF_TOP: lg_epilogue<EK_SDF_TOP, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_RELU: lg_epilogue<EK_RELU, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_SIGMOID: lg_epilogue<EK_SIGMOID, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_LINEAR_SIG: lg_epilogue<EK_LINEAR_SIG, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_RELIGHT_TOP: lg_epilogue<EK_RELIGHT_TOP, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_SWEEP: lg_epilogue<EK_SWEEP, NT>(g.E, acc, row_base, lane, g.P); break;
    case EK_VBACK: lg_epilogue<EK_VBACK, NT>(g.E, acc, row_base, lane, g.P); break;
    default: lg_epilogue<EK_RELU_MASK, NT>(g.E, acc, row_base, lane, g.P); break;
  }
}

template <int KIND, int NT>
__device__ __forceinline__ void lg_epilogue(const Epi& e0, const f32x16 (&acc)[NT], long row_base, int lane, long P) {
  Epi e = e0;
  e.kind = KIND;   // compile-time kind -> the switch in epi_apply folds away
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = row_base + (r & 3) + 8 * (r >> 2);
      if (row < P) epi_apply(e, row, col, acc[nt][r]);
    }
  }
}

template <int NT>
static void launch_layer_gemm(const LayerGemm& g, cnr_stream s) {
  const size_t lds = (size_t)(2 * LG_BM * LG_LD + 2 * NT * 32 * LG_LD) * sizeof(float);
  const unsigned grid = (unsigned)((g.P + LG_BM - 1) / LG_BM);
  if (grid == 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_gemm_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  TimingScope ts_("layer_gemm", 0, NT, g.P, g.N, g.K, 1, s);
  hipLaunchKernelGGL(layer_gemm_kernel<NT>, dim3(grid), dim3(256), lds, s, g);
}

void be_layer_gemm(const LayerGemm& g, cnr_stream s) {
  const int nt = (g.N + 31) / 32;
  switch (nt) {
    case 1: launch_layer_gemm<1>(g, s); break;
    case 2: launch_layer_gemm<2>(g, s); break;
    case 3: launch_layer_gemm<3>(g, s); break;
    case 4: launch_layer_gemm<4>(g, s); break;
    case 5: launch_layer_gemm<5>(g, s); break;
    case 6: launch_layer_gemm<6>(g, s); break;
    case 7: launch_layer_gemm<7>(g, s); break;
    case 8: launch_layer_gemm<8>(g, s); break;
    case 9: launch_layer_gemm<9>(g, s); break;
    case 10: launch_layer_gemm<10>(g, s); break;
    default:
      if (g_first_error == hipSuccess) { g_first_error = hipErrorInvalidValue; g_first_error_where = "layer_gemm: N > 320"; }
      return;
  }
  CNR_LAUNCH_CHECK("layer_gemm");
}

// ================================================================================================
// weight-gradient GEMM:  dW[n][k] = sum_pt X[pt][n] * Y[pt][k]
// 8 waves as WR x WC, each wave (MT*32) x (KT*32); block tile TN x TK; points streamed 16 at a time.
// ================================================================================================
constexpr int DW_BP = 16;

template <int WR, int WC, int MT, int KT>
__global__ __launch_bounds__(512) void dw_gemm_kernel(const DwGemm g, int n0, int k0) {
  constexpr int TN = WR * MT * 32, TK = WC * KT * 32;
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

  f32x16 acc[MT][KT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f4 rx[NX], ry[NY];
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.0f;
  const bool want_colsum = g.colsum != nullptr && k0 == 0;

  auto load_slab = [&](int s) {
    const int pair = s / nslab_pair;
    const long pbase = p_begin + (long)(s - pair * nslab_pair) * DW_BP;
    const View& X = g.X[pair];
    const View& Y = g.Y[pair];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int idx = tid + i * 512;
      if (DW_BP * TN / 4 % 512 == 0 || idx < DW_BP * TN / 4) {
        int pl = idx / (TN / 4), c4 = idx % (TN / 4);
        long pt = pbase + pl;
        rx[i] = pt < p_end ? view_eval4(X, pt, n0 + c4 * 4) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      int idx = tid + i * 512;
      if (DW_BP * TK / 4 % 512 == 0 || idx < DW_BP * TK / 4) {
        int pl = idx / (TK / 4), c4 = idx % (TK / 4);
        long pt = pbase + pl;
        ry[i] = pt < p_end ? view_eval4(Y, pt, k0 + c4 * 4) : zero4;
      }
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int idx = tid + i * 512;
      if (DW_BP * TN / 4 % 512 == 0 || idx < DW_BP * TN / 4)
        *reinterpret_cast<f4*>(Xs + buf * DW_BP * TN + idx * 4) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      int idx = tid + i * 512;
      if (DW_BP * TK / 4 % 512 == 0 || idx < DW_BP * TK / 4)
        *reinterpret_cast<f4*>(Ys + buf * DW_BP * TK + idx * 4) = ry[i];
    }
  };

  if (nslab > 0) {
    load_slab(0);
    store_slab(0);
  }
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) load_slab(s + 1);
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
    if (s + 1 < nslab) store_slab(buf ^ 1);
    __syncthreads();
  }

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

template <int WR, int WC, int MT, int KT>
static void launch_dw(const DwGemm& g, int n0, int k0, cnr_stream s) {
  constexpr int TN = WR * MT * 32, TK = WC * KT * 32;
  const size_t lds = (size_t)(2 * DW_BP * (TN + TK)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_gemm_kernel<WR, WC, MT, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  TimingScope ts_("dw_gemm", 1, WR * 1000 + WC * 100 + MT * 10 + KT, g.P, (g.N - n0) < WR * MT * 32 ? (g.N - n0) : WR * MT * 32, (g.K - k0) < WC * KT * 32 ? (g.K - k0) : WC * KT * 32, g.npairs, s);
  hipLaunchKernelGGL((dw_gemm_kernel<WR, WC, MT, KT>), dim3(g.nchunk), dim3(512), lds, s, g, n0, k0);
}

void be_dw_gemm(const DwGemm& g, cnr_stream s) {
  // tile the [Npad x ldk] output: 256x256 main tiles, 256x64 column tails, 32x256 row tails
  for (int n0 = 0; n0 < g.N; n0 += 256) {
    const int nrem = g.N - n0;
    for (int k0 = 0; k0 < g.K;) {
      const int krem = g.K - k0;
      if (nrem > 32) {
        if (krem > 64) { launch_dw<4, 2, 2, 4>(g, n0, k0, s); k0 += 256; }
        else { launch_dw<8, 1, 1, 2>(g, n0, k0, s); k0 += 64; }
      } else {
        launch_dw<1, 8, 1, 1>(g, n0, k0, s); k0 += 256;
      }
    }
  }
  CNR_LAUNCH_CHECK("dw_gemm");
}


}  // namespace cnr
