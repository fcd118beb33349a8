// Does a second staging register set (activation tiles fetched two ahead) help the weight-stationary f16x3 layer GEMM?
// Plain DIRECT -> bias+ReLU layer, 524288 x 256 x 256.  Build: hipcc --offload-arch=gfx950 -O3 ws_depth_probe.hip -o ws_depth_probe
#include <hip/hip_runtime.h>
#ifndef ABL_NAME
#define ABL_NAME "full"
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int KB = 16, TP = 32, ALD = 256 * 2 + 16, APLANE = TP * ALD, ABUF = 2 * APLANE + 256;
__device__ __forceinline__ float pow2_scale_for(float m) { if (!(m > 0.f)) return 1.0f; int e; frexpf(m, &e); return ldexpf(1.0f, 14 - e); }
__global__ void split_w(const float* W, _Float16* W1, _Float16* W2, float* wsi, int N, int K) {
  int n = blockIdx.x; float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 64) mx = fmaxf(mx, fabsf(W[(long)n * K + k]));
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  const float s = pow2_scale_for(mx);
  for (int k = threadIdx.x; k < K; k += 64) { float x = W[(long)n * K + k] * s; _Float16 h1 = (_Float16)x; W1[(long)n * K + k] = h1; W2[(long)n * K + k] = (_Float16)(x - (float)h1); }
  if (threadIdx.x == 0) wsi[n] = 1.0f / s;
}

template <int DEPTH>
__global__ __launch_bounds__(512, 1) void ws_gemm(const float* __restrict__ A, const _Float16* __restrict__ W1, const _Float16* __restrict__ W2,
                                                  const float* __restrict__ wsi, const float* __restrict__ bias, float* __restrict__ C, long P, int tpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 w1[KB], w2[KB];
  {
    const long off = (long)(wave * 32 + (lane & 31)) * 256 + (lane >> 5) * 8;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) { w1[kb] = *reinterpret_cast<const f16x8*>(W1 + off + kb * 16); w2[kb] = *reinterpret_cast<const f16x8*>(W2 + off + kb * 16); }
  }
  const float4 bias4 = *reinterpret_cast<const float4*>(bias + wave * 32 + (lane & 7) * 4);
  const float4 ws4 = *reinterpret_cast<const float4*>(wsi + wave * 32 + (lane & 7) * 4);
  const long tile0 = (long)blockIdx.x * tpw;
  const int srow = tid >> 4, sc4 = tid & 15;
  const bool late = wave >= 4;
  f4 ra[4], rb[4];
#ifdef ABL_NOLOAD
#define LOADT_COND(t_) ((t_) == tile0)
#else
#define LOADT_COND(t_) true
#endif
#ifdef ABL_ROWLOAD
#define LOADT(R_, t_) if (LOADT_COND(t_)) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { long row = ((t_) * TP) + wave * 4 + i; if (row >= P) row = P - 1; \
    R_[i] = *reinterpret_cast<const f4*>(A + row * 256 + lane * 4); } }
#else
#define LOADT(R_, t_) if (LOADT_COND(t_)) { long row = ((t_) * TP) + srow; if (row >= P) row = P - 1; const float* ap = A + row * 256 + sc4 * 4; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) R_[i] = *reinterpret_cast<const f4*>(ap + i * 64); }
#endif
#define STORET(R_, buf_) { float mx = 0.f; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) mx = fmaxf(fmaxf(fmaxf(fabsf(R_[i].x), fabsf(R_[i].y)), fmaxf(fabsf(R_[i].z), fabsf(R_[i].w))), mx); \
    _Pragma("unroll") for (int d = 8; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 16)); \
    const float sc = pow2_scale_for(mx); unsigned char* base = smem + (buf_) * ABUF + srow * ALD + sc4 * 8; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { f16x4 h1, h2; float x; \
      x = R_[i].x * sc; h1[0] = (_Float16)x; h2[0] = (_Float16)(x - (float)h1[0]); x = R_[i].y * sc; h1[1] = (_Float16)x; h2[1] = (_Float16)(x - (float)h1[1]); \
      x = R_[i].z * sc; h1[2] = (_Float16)x; h2[2] = (_Float16)(x - (float)h1[2]); x = R_[i].w * sc; h1[3] = (_Float16)x; h2[3] = (_Float16)(x - (float)h1[3]); \
      *reinterpret_cast<f16x4*>(base + i * 128) = h1; *reinterpret_cast<f16x4*>(base + APLANE + i * 128) = h2; } \
    if (sc4 == 0) reinterpret_cast<float*>(smem + (buf_) * ABUF + 2 * APLANE)[srow] = 1.0f / sc; }
#ifdef ABL_NOMFMA
#define MFMA3(a1, a2, kb) if ((kb) == 0) { acc[0] += (float)a1[0] + (float)a2[0] + (float)w1[15][0] + (float)w2[15][0]; }
#else
#define MFMA3(a1, a2, kb) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0);
#endif
#ifdef ABL_NOFRAG
#define FRAGKB(kb) ((kb) & 1)
#else
#define FRAGKB(kb) (kb)
#endif
#ifdef ABL_NOSTORE
#define STORE_COND(v) ((v).x == 12345.678f)
#else
#define STORE_COND(v) true
#endif
#ifdef ABL_TWOACC
#define COMPUTE(t_, buf_) { f32x16 acc, accx; _Pragma("unroll") for (int j = 0; j < 16; ++j) { acc[j] = 0.f; accx[j] = 0.f; } \
    const unsigned char* Ab = smem + (buf_) * ABUF + (lane & 31) * ALD + (lane >> 5) * 16; \
    _Pragma("unroll") for (int kb = 0; kb < KB; kb += 2) { const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32); const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + APLANE + kb * 32); \
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(Ab + (kb + 1) * 32); const f16x8 b2 = *reinterpret_cast<const f16x8*>(Ab + APLANE + (kb + 1) * 32); \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0); accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, w2[kb + 1], accx, 0, 0, 0); \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0); accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, w1[kb + 1], accx, 0, 0, 0); \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0); accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, w1[kb + 1], accx, 0, 0, 0); } \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) acc[j] += accx[j]; \
    COMPUTE_TAIL(t_, buf_) }
#else
#define COMPUTE(t_, buf_) { f32x16 acc; _Pragma("unroll") for (int j = 0; j < 16; ++j) acc[j] = 0.f; \
    const unsigned char* Ab = smem + (buf_) * ABUF + (lane & 31) * ALD + (lane >> 5) * 16; \
    _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) { const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + FRAGKB(kb) * 32); const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + APLANE + FRAGKB(kb) * 32); \
      MFMA3(a1, a2, kb) } \
    COMPUTE_TAIL(t_, buf_) }
#endif
#define COMPUTE_TAIL(t_, buf_) \
    const float* rs = reinterpret_cast<const float*>(smem + (buf_) * ABUF + 2 * APLANE); const int hi = lane >> 5, cl = lane & 31; \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + cl] = acc[r]; \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { const int rr = (lane >> 3) + 8 * i, cc = lane & 7; const long row = (t_) * TP + rr; const float rsc = rs[rr]; \
      f4 v = *reinterpret_cast<const f4*>(T + rr * 36 + cc * 4); \
      v.x = fmaxf(v.x * (rsc * ws4.x) + bias4.x, 0.f); v.y = fmaxf(v.y * (rsc * ws4.y) + bias4.y, 0.f); v.z = fmaxf(v.z * (rsc * ws4.z) + bias4.z, 0.f); v.w = fmaxf(v.w * (rsc * ws4.w) + bias4.w, 0.f); \
      if (row < P && STORE_COND(v)) *reinterpret_cast<f4*>(C + row * 256 + wave * 32 + cc * 4) = v; } \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
  float* T = reinterpret_cast<float*>(smem + 2 * ABUF) + wave * (32 * 36);
  const long t1 = tile0 + tpw;
  LOADT(ra, tile0) STORET(ra, 0)
  if (DEPTH == 1) {
    if (late && tile0 + 1 < t1) LOADT(ra, tile0 + 1)
    __syncthreads();
    for (long t = tile0; t < t1; ++t) {
      const int buf = (int)((t - tile0) & 1); const bool more = t + 1 < t1;
      if (!late) { if (more) LOADT(ra, t + 1) } else if (more) { STORET(ra, buf ^ 1) if (t + 2 < t1) LOADT(ra, t + 2) }
      COMPUTE(t, buf)
      if (!late && more) STORET(ra, buf ^ 1)
      __syncthreads();
    }
  } else {
    // two register sets: early waves keep tiles t+1 (consumed at the end of iteration t) and t+2 in flight, late waves t+2 and t+3
    if (tile0 + 1 < t1) LOADT(rb, tile0 + 1)            // set b: odd tiles (relative), set a: even
    if (late && tile0 + 2 < t1) LOADT(ra, tile0 + 2)
    __syncthreads();
    for (long t = tile0; t < t1; t += 2) {
      {   // even relative tile t: next tile t+1 lives in rb
        const bool more = t + 1 < t1;
        if (!late) { if (t + 2 < t1) LOADT(ra, t + 2) } else if (more) { STORET(rb, 1) if (t + 3 < t1) LOADT(rb, t + 3) }
        COMPUTE(t, 0)
        if (!late && more) STORET(rb, 1)
        __syncthreads();
      }
      if (t + 1 < t1) {   // odd relative tile t+1: next tile t+2 lives in ra
        const bool more = t + 2 < t1;
        if (!late) { if (t + 3 < t1) LOADT(rb, t + 3) } else if (more) { STORET(ra, 0) if (t + 4 < t1) LOADT(ra, t + 4) }
        COMPUTE(t + 1, 1)
        if (!late && more) STORET(ra, 0)
        __syncthreads();
      }
    }
  }
}

int main() {
  const long P = 524288; const int K = 256, N = 256;
  std::vector<float> hA((size_t)P * K), hW((size_t)N * K), hb(N);
  srand(1);
  for (auto& x : hA) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hW) x = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
  for (auto& x : hb) x = (rand() / (float)RAND_MAX) * 0.1f;
  float *A, *W, *b, *C, *wsi; _Float16 *H1, *H2;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&C, (size_t)P * N * 4));
  CK(hipMalloc(&H1, hW.size() * 2)); CK(hipMalloc(&H2, hW.size() * 2)); CK(hipMalloc(&wsi, N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_w, dim3(N), dim3(64), 0, 0, W, H1, H2, wsi, N, K);
  const long ntiles = P / TP; const int nwg = 256; const int tpw = (int)(ntiles / nwg);
  const size_t lds = (size_t)2 * ABUF + 8 * 32 * 36 * 4;
  for (int depth = 2; depth <= 2; ++depth) {
    auto kern = depth == 1 ? ws_gemm<1> : ws_gemm<2>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, 0, A, H1, H2, wsi, b, C, P, tpw);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, 0, A, H1, H2, wsi, b, C, P, tpw);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; CK(hipGetLastError());
    std::vector<float> hC(256 * (size_t)N);
    CK(hipMemcpy(hC.data(), C + (size_t)(P - 256) * N, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int r = 0; r < 256; r += 7) for (int n = 0; n < N; n += 5) { double s = hb[n]; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(P - 256 + r) * K + k] * hW[(size_t)n * K + k]; if (s < 0) s = 0; maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n])); }
    printf(ABL_NAME " depth %d: %.3f ms  %.0f GB/s  maxerr %.2e\n", depth, ms, 2.0 * P * 1024 / (ms * 1e-3) / 1e9, maxerr);
  }
  return 0;
}
