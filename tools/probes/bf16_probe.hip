// Probe: fp32 GEMM emulated with a 3-way bf16 split (a = a1 + a2 + a3, 8 significand bits each) and six
// v_mfma_f32_32x32x16_bf16 per product: a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (terms below 2^-24 dropped).
// Build: hipcc --offload-arch=gfx950 -O3 bf16_probe.hip -o bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned short bf16_rn(float x) {   // round-to-nearest-even fp32 -> bf16 bits
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h1, unsigned short& h2, unsigned short& h3) {
  h1 = bf16_rn(x); float r = x - bf16_f(h1);
  h2 = bf16_rn(r); r = r - bf16_f(h2);
  h3 = bf16_rn(r);
}

// pre-split weights: W[N][K] fp32 -> 3 planes of bf16 [N][K]
__global__ void split_weights(const float* W, unsigned short* W1, unsigned short* W2, unsigned short* W3, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { unsigned short a, b, c; split3(W[i], a, b, c); W1[i] = a; W2[i] = b; W3[i] = c; }
}

// C[P][N] = relu(A*W^T + b): 128 x (NT*32) tile, BK = 32 per slab (two k16 MFMA blocks), single LDS buffer + register prefetch
template <int NT, int NTERMS, int BK, int DBUF, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_bf16x(const float* __restrict__ A, const unsigned short* __restrict__ W1,
                                                      const unsigned short* __restrict__ W2, const unsigned short* __restrict__ W3,
                                                      const float* __restrict__ bias, float* __restrict__ C, long P, int K, int N) {
  constexpr int LDB = BK * 2 + 16;   // bytes per LDS row (+16 pad)
  constexpr int ABUF = 3 * 128 * LDB, BBUF = 3 * NT * 32 * LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As0 = smem;                                // [DBUF+1][3][128][LDB]
  unsigned char* Bs0 = smem + (DBUF + 1) * ABUF;            // [DBUF+1][3][NT*32][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * 128;
  const int nslab = K / BK;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  // A staging: 128 rows x 32 floats = 1024 float4 -> 4 per thread; W staging: 3 planes x 256 rows x 32 bf16 (64 B = 4 x 16 B) = 3072 x 16 B -> 12 per thread
  constexpr int NA = 128 * BK / 4 / 256;          // float4 of A per thread per slab
  constexpr int NW = NT * 32 * BK / 8 / 256;      // 16-byte pieces of one W plane per thread per slab
  f4 ra[NA];
  u32x4 rw[3 * NW];
#define LOAD_SLAB(s_)                                                                                         \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                          \
      int idx = tid + i * 256; int r = idx / (BK / 4), c4 = idx % (BK / 4);                                   \
      long row = row0 + r; if (row >= P) row = P - 1;                                                         \
      ra[i] = *reinterpret_cast<const f4*>(A + row * K + (s_) * BK + c4 * 4);                                 \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                          \
      int idx = tid + i * 256; int r = idx / (BK / 8), c8 = idx % (BK / 8);                                   \
      long off = (long)r * K + (s_) * BK + c8 * 8;                                                            \
      rw[i] = *reinterpret_cast<const u32x4*>(W1 + off);                                                      \
      rw[NW + i] = *reinterpret_cast<const u32x4*>(W2 + off);                                                 \
      rw[2 * NW + i] = *reinterpret_cast<const u32x4*>(W3 + off);                                             \
    }                                                                                                         \
  }
#define STORE_SLAB(buf_)                                                                                      \
  {                                                                                                           \
    unsigned char* As = As0 + (buf_) * ABUF; unsigned char* Bs = Bs0 + (buf_) * BBUF;                         \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                          \
      int idx = tid + i * 256; int r = idx / (BK / 4), c4 = idx % (BK / 4);                                   \
      unsigned short h1[4], h2[4], h3[4];                                                                     \
      split3(ra[i].x, h1[0], h2[0], h3[0]); split3(ra[i].y, h1[1], h2[1], h3[1]);                             \
      split3(ra[i].z, h1[2], h2[2], h3[2]); split3(ra[i].w, h1[3], h2[3], h3[3]);                             \
      uint2 p1 = make_uint2(h1[0] | (h1[1] << 16), h1[2] | (h1[3] << 16));                                    \
      uint2 p2 = make_uint2(h2[0] | (h2[1] << 16), h2[2] | (h2[3] << 16));                                    \
      uint2 p3 = make_uint2(h3[0] | (h3[1] << 16), h3[2] | (h3[3] << 16));                                    \
      *reinterpret_cast<uint2*>(As + (0 * 128 + r) * LDB + c4 * 8) = p1;                                      \
      *reinterpret_cast<uint2*>(As + (1 * 128 + r) * LDB + c4 * 8) = p2;                                      \
      *reinterpret_cast<uint2*>(As + (2 * 128 + r) * LDB + c4 * 8) = p3;                                      \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                          \
      int idx = tid + i * 256; int r = idx / (BK / 8), c8 = idx % (BK / 8);                                   \
      *reinterpret_cast<u32x4*>(Bs + (0 * NT * 32 + r) * LDB + c8 * 16) = rw[i];                              \
      *reinterpret_cast<u32x4*>(Bs + (1 * NT * 32 + r) * LDB + c8 * 16) = rw[NW + i];                         \
      *reinterpret_cast<u32x4*>(Bs + (2 * NT * 32 + r) * LDB + c8 * 16) = rw[2 * NW + i];                     \
    }                                                                                                         \
  }
  LOAD_SLAB(0)
  if (DBUF) { STORE_SLAB(0) __syncthreads(); }
  for (int s = 0; s < nslab; ++s) {
    const int buf = DBUF ? (s & 1) : 0;
    if (!DBUF) {
      __syncthreads();            // previous slab fully consumed
      STORE_SLAB(0)
      __syncthreads();
    }
    if (s + 1 < nslab) LOAD_SLAB(s + 1)
    const unsigned char* Ab = As0 + buf * ABUF + (wave * 32 + (lane & 31)) * LDB + (lane >> 5) * 16;
    const unsigned char* Bb = Bs0 + buf * BBUF + (lane & 31) * LDB + (lane >> 5) * 16;
#pragma unroll
    for (int kb = 0; kb < BK / 16; ++kb) {
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + 0 * 128 * LDB + kb * 32);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(Ab + 1 * 128 * LDB + kb * 32);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(Ab + 2 * 128 * LDB + kb * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bb + (0 * NT * 32 + nt * 32) * LDB + kb * 32);
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(Bb + (1 * NT * 32 + nt * 32) * LDB + kb * 32);
        if (NTERMS >= 6) {
          const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(Bb + (2 * NT * 32 + nt * 32) * LDB + kb * 32);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[nt], 0, 0, 0);
        }
        if (NTERMS >= 3) {
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[nt], 0, 0, 0);
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[nt], 0, 0, 0);
      }
    }
    if (DBUF) {
      if (s + 1 < nslab) STORE_SLAB(buf ^ 1)
      __syncthreads();
    }
  }
  __syncthreads();
  float* T = reinterpret_cast<float*>(smem) + wave * (32 * 33);
  const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + cl] = acc[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i, c4 = lane & 7;
      const long row = row0 + wave * 32 + rr;
      const int col = nt * 32 + c4 * 4;
      f4 v;
      v.x = fmaxf(T[rr * 33 + c4 * 4 + 0] + bias[col + 0], 0.f); v.y = fmaxf(T[rr * 33 + c4 * 4 + 1] + bias[col + 1], 0.f);
      v.z = fmaxf(T[rr * 33 + c4 * 4 + 2] + bias[col + 2], 0.f); v.w = fmaxf(T[rr * 33 + c4 * 4 + 3] + bias[col + 3], 0.f);
      if (row < P) *reinterpret_cast<f4*>(C + row * N + col) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
  const unsigned grid = (unsigned)((P + 127) / 128);
  const double flop = 2.0 * P * N * K;
  auto run = [&](const char* name, auto kernel, int BK, int DBUF) {
    size_t lds = (size_t)(DBUF + 1) * (3 * 128 + 3 * 256) * (BK * 2 + 16);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(C, 0, (size_t)P * N * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, A, W1, W2, W3, b, C, P, K, N);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, A, W1, W2, W3, b, C, P, K, N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    CK(hipGetLastError());
    std::vector<float> hC(256 * (size_t)N);
    CK(hipMemcpy(hC.data(), C + (size_t)(P - 256) * N, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0, err32 = 0;
    for (int r = 0; r < 256; r += 5) for (int n = 0; n < N; n += 3) {
      double s = hb[n]; float s32 = 0.f;
      for (int k = 0; k < K; ++k) { s += (double)hA[(size_t)(P - 256 + r) * K + k] * hW[(size_t)n * K + k]; s32 = fmaf(hA[(size_t)(P - 256 + r) * K + k], hW[(size_t)n * K + k], s32); }
      s32 += hb[n];
      double s0 = s; if (s < 0) s = 0;
      maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n])); maxref = fmax(maxref, fabs(s));
      err32 = fmax(err32, fabs(s0 - s32));
    }
    printf("%-22s %.3f ms  %.1f TF/s-equivalent   maxerr vs f64 %.2e (fp32 fma chain: %.2e, max|ref| %.2f)\n", name, ms, flop / (ms * 1e-3) / 1e12, maxerr, err32, maxref);
  };
  run("x6 BK32 single 1wg", gemm_bf16x<8, 6, 32, 0, 1>, 32, 0);
  run("x6 BK16 single 2wg", gemm_bf16x<8, 6, 16, 0, 2>, 16, 0);
  run("x6 BK16 double 1wg", gemm_bf16x<8, 6, 16, 1, 1>, 16, 1);
  run("x1 BK16 single 2wg", gemm_bf16x<8, 1, 16, 0, 2>, 16, 0);
  run("x3 BK16 single 2wg", gemm_bf16x<8, 3, 16, 0, 2>, 16, 0);
  return 0;
}
