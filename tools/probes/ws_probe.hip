// Probe: WEIGHT-STATIONARY fp32-accurate GEMM for the 256x256 MLP layers.
// The whole layer (256 x 256 weights) lives in the REGISTER FILE of one CU: 8 waves x 32 output columns, each wave keeps its
// slice of W as pre-split bf16 planes in MFMA B-operand layout (3 planes x 16 k-blocks x 4 VGPRs = 192 VGPRs).  Points stream
// through: a 32-point tile is staged (prologue math + 3-way bf16 split) into LDS once and read by all 8 waves.  No weight traffic
// at all after the prologue, 16 accumulator registers per wave, six v_mfma_f32_32x32x16_bf16 per product (a1b1+a1b2+a2b1+a1b3+a2b2+a3b1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned int bf16_rn_bits(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned int& h1, unsigned int& h2, unsigned int& h3) {
  h1 = bf16_rn_bits(x); float r = x - __uint_as_float(h1 << 16);
  h2 = bf16_rn_bits(r); r = r - __uint_as_float(h2 << 16);
  h3 = bf16_rn_bits(r);
}
__global__ void split_weights(const float* W, unsigned short* W1, unsigned short* W2, unsigned short* W3, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { unsigned int a, b, c; split3(W[i], a, b, c); W1[i] = a; W2[i] = b; W3[i] = c; }
}

constexpr int KB = 16;                 // k16 blocks (K = 256)
constexpr int TP = 32;                 // points per tile
constexpr int ALD = 256 * 2 + 16;      // bytes per LDS row of one activation plane (528)
constexpr int APLANE = TP * ALD;       // 16896 B
constexpr int ABUF = 3 * APLANE;       // 50688 B per buffer

template <int NTERMS>
__global__ __launch_bounds__(512, 1) void ws_gemm(const float* __restrict__ A, const unsigned short* __restrict__ W1,
                                                  const unsigned short* __restrict__ W2, const unsigned short* __restrict__ W3,
                                                  const float* __restrict__ bias, float* __restrict__ C, long P, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- resident weights: this wave's 32 output columns, all K, three planes
  bf16x8 w1[KB], w2[KB], w3[KB];
  {
    const long off = (long)(wave * 32 + (lane & 31)) * 256 + (lane >> 5) * 8;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      w1[kb] = *reinterpret_cast<const bf16x8*>(W1 + off + kb * 16);
      w2[kb] = *reinterpret_cast<const bf16x8*>(W2 + off + kb * 16);
      w3[kb] = *reinterpret_cast<const bf16x8*>(W3 + off + kb * 16);
    }
  }
  const float4 bias4 = *reinterpret_cast<const float4*>(bias + wave * 32 + (lane & 7) * 4);
  const long tile0 = (long)blockIdx.x * tiles_per_wg;
  const int srow = tid >> 4, sc4 = tid & 15;          // staging: 16 threads per row, 4 passes of 64 floats
  f4 ra[4];
#define WS_LOAD(t_)                                                                     \
  {                                                                                     \
    long row = ((t_) * TP) + srow; if (row >= P) row = P - 1;                            \
    const float* ap = A + row * 256 + sc4 * 4;                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f4*>(ap + i * 64); \
  }
#define WS_STORE(buf_)                                                                  \
  {                                                                                     \
    unsigned char* base = smem + (buf_) * ABUF + srow * ALD + sc4 * 8;                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
      unsigned int a[4], b[4], c[4];                                                    \
      split3(ra[i].x, a[0], b[0], c[0]); split3(ra[i].y, a[1], b[1], c[1]);             \
      split3(ra[i].z, a[2], b[2], c[2]); split3(ra[i].w, a[3], b[3], c[3]);             \
      u32x2 p1 = {a[0] | (a[1] << 16), a[2] | (a[3] << 16)};                            \
      u32x2 p2 = {b[0] | (b[1] << 16), b[2] | (b[3] << 16)};                            \
      u32x2 p3 = {c[0] | (c[1] << 16), c[2] | (c[3] << 16)};                            \
      *reinterpret_cast<u32x2*>(base + 0 * APLANE + i * 128) = p1;                      \
      *reinterpret_cast<u32x2*>(base + 1 * APLANE + i * 128) = p2;                      \
      *reinterpret_cast<u32x2*>(base + 2 * APLANE + i * 128) = p3;                      \
    }                                                                                   \
  }
  WS_LOAD(tile0)
  WS_STORE(0)
  __syncthreads();
  float* T = reinterpret_cast<float*>(smem + 2 * ABUF) + wave * (32 * 36);
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int buf = t & 1;
    const long tile = tile0 + t;
    if (t + 1 < tiles_per_wg) WS_LOAD(tile + 1)
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const unsigned char* Ab = smem + buf * ABUF + (lane & 31) * ALD + (lane >> 5) * 16;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + 0 * APLANE + kb * 32);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(Ab + 1 * APLANE + kb * 32);
      if (NTERMS >= 6) {
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(Ab + 2 * APLANE + kb * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w3[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, w1[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w2[kb], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w2[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w1[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1[kb], acc, 0, 0, 0);
    }
    // epilogue of this 32 x 32 tile: transpose through the wave's private LDS tile, 4 columns per lane
    {
      const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + cl] = acc[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = (lane >> 3) + 8 * i, cc = lane & 7;
        const long row = tile * TP + rr;
        f4 v = *reinterpret_cast<const f4*>(T + rr * 36 + cc * 4);
        v.x = fmaxf(v.x + bias4.x, 0.f); v.y = fmaxf(v.y + bias4.y, 0.f); v.z = fmaxf(v.z + bias4.z, 0.f); v.w = fmaxf(v.w + bias4.w, 0.f);
        if (row < P) *reinterpret_cast<f4*>(C + row * 256 + wave * 32 + cc * 4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (t + 1 < tiles_per_wg) WS_STORE(buf ^ 1)
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// f16 x 3 with exact power-of-two row / column scaling: a = (a1 + a2) * 2^ea (11 + 11 significand bits), w likewise per column;
// product = a1w1 + a1w2 + a2w1 (dropped a2w2 < 2^-22); 3 MFMAs per k16 block, 2 planes -> 128 weight VGPRs.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float pow2_scale_for(float maxabs) {   // power of two s such that maxabs * s is in [2^13, 2^14)
  if (!(maxabs > 0.f)) return 1.0f;
  int e; frexpf(maxabs, &e);            // maxabs = m * 2^e, m in [0.5, 1)
  return ldexpf(1.0f, 14 - e);
}
__global__ void split_weights_f16(const float* W, _Float16* W1, _Float16* W2, float* wscale_inv, int N, int K) {
  int n = blockIdx.x;                    // one block (64 threads) per output row
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += 64) mx = fmaxf(mx, fabsf(W[(long)n * K + k]));
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  const float s = pow2_scale_for(mx);
  for (int k = threadIdx.x; k < K; k += 64) {
    float x = W[(long)n * K + k] * s;
    _Float16 h1 = (_Float16)x; _Float16 h2 = (_Float16)(x - (float)h1);
    W1[(long)n * K + k] = h1; W2[(long)n * K + k] = h2;
  }
  if (threadIdx.x == 0) wscale_inv[n] = 1.0f / s;
}

constexpr int ALD16 = 256 * 2 + 16;
constexpr int APLANE16 = TP * ALD16;
constexpr int ABUF16 = 2 * APLANE16 + 256;   // 2 planes + 32 row scales (padded)

__global__ __launch_bounds__(512, 1) void ws_gemm_f16(const float* __restrict__ A, const _Float16* __restrict__ W1, const _Float16* __restrict__ W2,
                                                      const float* __restrict__ wscale_inv, const float* __restrict__ bias, float* __restrict__ C,
                                                      long P, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 w1[KB], w2[KB];
  {
    const long off = (long)(wave * 32 + (lane & 31)) * 256 + (lane >> 5) * 8;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      w1[kb] = *reinterpret_cast<const f16x8*>(W1 + off + kb * 16);
      w2[kb] = *reinterpret_cast<const f16x8*>(W2 + off + kb * 16);
    }
  }
  const float4 bias4 = *reinterpret_cast<const float4*>(bias + wave * 32 + (lane & 7) * 4);
  const float4 ws4 = *reinterpret_cast<const float4*>(wscale_inv + wave * 32 + (lane & 7) * 4);
  const long tile0 = (long)blockIdx.x * tiles_per_wg;
  const int srow = tid >> 4, sc4 = tid & 15;
  f4 ra[4];
#define F_LOAD(t_)                                                                      \
  {                                                                                     \
    long row = ((t_) * TP) + srow; if (row >= P) row = P - 1;                            \
    const float* ap = A + row * 256 + sc4 * 4;                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f4*>(ap + i * 64); \
  }
#define F_STORE(buf_)                                                                   \
  {                                                                                     \
    float mx = 0.f;                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) mx = fmaxf(fmaxf(fmaxf(fabsf(ra[i].x), fabsf(ra[i].y)), fmaxf(fabsf(ra[i].z), fabsf(ra[i].w))), mx); \
    _Pragma("unroll") for (int d = 8; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 16)); \
    const float sc = pow2_scale_for(mx);                                                \
    unsigned char* base = smem + (buf_) * ABUF16 + srow * ALD16 + sc4 * 8;               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
      f16x4 h1, h2;                                                                     \
      float x;                                                                          \
      x = ra[i].x * sc; h1[0] = (_Float16)x; h2[0] = (_Float16)(x - (float)h1[0]);      \
      x = ra[i].y * sc; h1[1] = (_Float16)x; h2[1] = (_Float16)(x - (float)h1[1]);      \
      x = ra[i].z * sc; h1[2] = (_Float16)x; h2[2] = (_Float16)(x - (float)h1[2]);      \
      x = ra[i].w * sc; h1[3] = (_Float16)x; h2[3] = (_Float16)(x - (float)h1[3]);      \
      *reinterpret_cast<f16x4*>(base + i * 128) = h1;                                   \
      *reinterpret_cast<f16x4*>(base + APLANE16 + i * 128) = h2;                        \
    }                                                                                   \
    if (sc4 == 0) reinterpret_cast<float*>(smem + (buf_) * ABUF16 + 2 * APLANE16)[srow] = 1.0f / sc; \
  }
  F_LOAD(tile0)
  F_STORE(0)
  __syncthreads();
  float* T = reinterpret_cast<float*>(smem + 2 * ABUF16) + wave * (32 * 36);
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int buf = t & 1;
    const long tile = tile0 + t;
    if (t + 1 < tiles_per_wg) F_LOAD(tile + 1)
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const unsigned char* Ab = smem + buf * ABUF16 + (lane & 31) * ALD16 + (lane >> 5) * 16;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32);
      const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + APLANE16 + kb * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0);
    }
    {
      const float* rs = reinterpret_cast<const float*>(smem + buf * ABUF16 + 2 * APLANE16);
      const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + cl] = acc[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = (lane >> 3) + 8 * i, cc = lane & 7;
        const long row = tile * TP + rr;
        const float rsc = rs[rr];
        f4 v = *reinterpret_cast<const f4*>(T + rr * 36 + cc * 4);
        v.x = fmaxf(v.x * (rsc * ws4.x) + bias4.x, 0.f); v.y = fmaxf(v.y * (rsc * ws4.y) + bias4.y, 0.f);
        v.z = fmaxf(v.z * (rsc * ws4.z) + bias4.z, 0.f); v.w = fmaxf(v.w * (rsc * ws4.w) + bias4.w, 0.f);
        if (row < P) *reinterpret_cast<f4*>(C + row * 256 + wave * 32 + cc * 4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (t + 1 < tiles_per_wg) F_STORE(buf ^ 1)
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const long P = argc > 1 ? atol(argv[1]) : 524288;
  const int K = 256, N = 256;
  std::vector<float> hA((size_t)P * K), hW((size_t)N * K), hb(N);
  srand(1);
  for (auto& x : hA) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hW) x = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
  for (auto& x : hb) x = (rand() / (float)RAND_MAX) * 0.1f;
  float *A, *W, *b, *C; unsigned short *W1, *W2, *W3;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&C, (size_t)P * N * 4));
  CK(hipMalloc(&W1, hW.size() * 2)); CK(hipMalloc(&W2, hW.size() * 2)); CK(hipMalloc(&W3, hW.size() * 2));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_weights, dim3((N * K + 255) / 256), dim3(256), 0, 0, W, W1, W2, W3, (long)N * K);
  const double flop = 2.0 * P * N * K;
  const long ntiles = (P + TP - 1) / TP;
  for (int variant = 0; variant < 4; ++variant) {
    const int nwg = variant < 2 ? 256 : (variant == 2 ? 512 : 1024);
    const int tpw = (int)((ntiles + nwg - 1) / nwg);
    auto kernel = variant == 1 ? ws_gemm<3> : ws_gemm<6>;
    size_t lds = (size_t)2 * ABUF + 8 * 32 * 36 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(C, 0, (size_t)P * N * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), lds, 0, A, W1, W2, W3, b, C, P, tpw);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), lds, 0, A, W1, W2, W3, b, C, P, tpw);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    CK(hipGetLastError());
    std::vector<float> hC(256 * (size_t)N);
    CK(hipMemcpy(hC.data(), C + (size_t)(P - 256) * N, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int r = 0; r < 256; r += 5) for (int n = 0; n < N; n += 3) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(P - 256 + r) * K + k] * hW[(size_t)n * K + k];
      if (s < 0) s = 0;
      maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n]));
    }
    printf("weight-stationary bf16x%d, %4d WGs: %.3f ms  %.1f TF/s-equivalent  maxerr %.2e  lds %zu\n", variant == 1 ? 3 : 6, nwg, ms, flop / (ms * 1e-3) / 1e12, maxerr, lds);
  }

  {   // f16 x 3 variant, also on a copy of A whose rows span 12 orders of magnitude (cotangent-like data)
    _Float16 *H1, *H2; float* wsi;
    CK(hipMalloc(&H1, hW.size() * 2)); CK(hipMalloc(&H2, hW.size() * 2)); CK(hipMalloc(&wsi, N * 4));
    hipLaunchKernelGGL(split_weights_f16, dim3(N), dim3(64), 0, 0, W, H1, H2, wsi, N, K);
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
        for (long r = 0; r < P; ++r) { float sc = powf(10.f, -12.f * (float)(r % 97) / 96.f); for (int k = 0; k < K; ++k) hA[(size_t)r * K + k] *= sc * ((k % 7 == 0) ? 1e-3f : 1.f); }
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        for (auto& x : hb) x = 0.f;
        CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
      }
      const int nwg = 512; const int tpw = (int)((ntiles + nwg - 1) / nwg);
      size_t lds = (size_t)2 * ABUF16 + 8 * 32 * 36 * 4;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ws_gemm_f16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ws_gemm_f16, dim3(nwg), dim3(512), lds, 0, A, H1, H2, wsi, b, C, P, tpw);
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(ws_gemm_f16, dim3(nwg), dim3(512), lds, 0, A, H1, H2, wsi, b, C, P, tpw);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      CK(hipGetLastError());
      std::vector<float> hC(256 * (size_t)N);
      CK(hipMemcpy(hC.data(), C + (size_t)(P - 256) * N, hC.size() * 4, hipMemcpyDeviceToHost));
      double maxrel = 0, maxrel32 = 0;
      for (int r = 0; r < 256; r += 3) {
        double rowscale = 0; for (int k = 0; k < K; ++k) rowscale = fmax(rowscale, fabs(hA[(size_t)(P - 256 + r) * K + k]));
        for (int n = 0; n < N; n += 3) {
          double s = 0; float s32 = 0.f;
          for (int k = 0; k < K; ++k) { s += (double)hA[(size_t)(P - 256 + r) * K + k] * hW[(size_t)n * K + k]; s32 = fmaf(hA[(size_t)(P - 256 + r) * K + k], hW[(size_t)n * K + k], s32); }
          double got = hC[(size_t)r * N + n]; double ref = s + hb[n]; if (ref < 0) ref = 0;
          if (pass == 0) { maxrel = fmax(maxrel, fabs(got - ref)); maxrel32 = fmax(maxrel32, fabs((double)s32 - s)); }
          else if (s > 0) { maxrel = fmax(maxrel, fabs(got - s) / (rowscale * 0.1 * 16)); maxrel32 = fmax(maxrel32, fabs((double)s32 - s) / (rowscale * 0.1 * 16)); }
        }
      }
      printf("weight-stationary f16x3 (row/col pow2 scaling) %s: %.3f ms  %.1f TF/s-equivalent  err %.2e (fp32 fma chain %.2e)\n", pass == 0 ? "uniform data" : "wide-range rows (rel to row scale)", ms, flop / (ms * 1e-3) / 1e12, maxrel, maxrel32);
    }
  }
  return 0;
}