// Stand-alone probe: how fast can the 128 x (NT*32) FP32-MFMA layer-GEMM tiling go with a clean inner loop?
// Variants differ in K-slab width and in how the staging is scheduled.  Build: hipcc --offload-arch=gfx950 -O3 gemm_probe.hip -o gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define LOAD_SLAB(s_)                                                                            \
  {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                               \
      int idx = tid + i * 256; int r = idx / F4R, c4 = idx % F4R;                                  \
      long row = row0 + r; if (row >= P) row = P - 1;                                              \
      ra[i] = *reinterpret_cast<const f4*>(A + row * lda + (s_) * BK + c4 * 4);                    \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                               \
      int idx = tid + i * 256; int r = idx / F4R, c4 = idx % F4R;                                  \
      rb[i] = *reinterpret_cast<const f4*>(W + (long)r * ldw + (s_) * BK + c4 * 4);                \
    }                                                                                              \
  }
#define STORE_SLAB(buf_)                                                                           \
  {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                               \
      int idx = tid + i * 256; int r = idx / F4R, c4 = idx % F4R;                                  \
      *reinterpret_cast<f4*>(As + ((buf_) * 128 + r) * LD + c4 * 4) = ra[i];                       \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                               \
      int idx = tid + i * 256; int r = idx / F4R, c4 = idx % F4R;                                  \
      *reinterpret_cast<f4*>(Bs + ((buf_) * NT * 32 + r) * LD + c4 * 4) = rb[i];                   \
    }                                                                                              \
  }

// C[P][N] = relu(A[P][K] * W[N][K]^T + b)
template <int NT, int BK, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_v1(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                     float* __restrict__ C, long P, int K, int lda, int ldw, int ldc) {
  constexpr int LD = BK + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * 128 * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * 128;
  const int nslab = K / BK;
  constexpr int F4R = BK / 4;                       // float4 per row per slab
  constexpr int NA = 128 * F4R / 256, NB = NT * 32 * F4R / 256;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f4 ra[NA], rb[NB];
  LOAD_SLAB(0) STORE_SLAB(0) __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) LOAD_SLAB(s + 1)
    const float* Ab = As + (buf * 128 + wave * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* Bb = Bs + (buf * NT * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const f4 a = *reinterpret_cast<const f4*>(Ab + kb * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4*>(Bb + nt * 32 * LD + kb * 8);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
      }
    }
    if (s + 1 < nslab) STORE_SLAB(buf ^ 1)
    __syncthreads();
  }
  const long rb0 = row0 + wave * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 32 + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = rb0 + (r & 3) + 8 * (r >> 2);
      if (row < P) { float v = acc[nt][r] + bv; C[row * ldc + col] = v > 0.f ? v : 0.f; }
    }
  }
}

// v2: same tiling, epilogue transposed through LDS so that every lane stores 16 B (float4) and a wave writes whole 512 B rows
template <int NT, int BK, int MINW, int ABL>
__global__ __launch_bounds__(256, MINW) void gemm_v2(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                     float* __restrict__ C, long P, int K, int lda, int ldw, int ldc) {
  constexpr int LD = BK + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * 128 * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * 128;
  const int nslab = K / BK;
  constexpr int F4R = BK / 4;
  constexpr int NA = 128 * F4R / 256, NB = NT * 32 * F4R / 256;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f4 ra[NA], rb[NB];
  LOAD_SLAB(0) STORE_SLAB(0) __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (ABL == 0 && s + 1 < nslab) LOAD_SLAB(s + 1)
    const float* Ab = As + (buf * 128 + wave * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* Bb = Bs + (buf * NT * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const f4 a = *reinterpret_cast<const f4*>(Ab + kb * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4*>(Bb + nt * 32 * LD + kb * 8);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
      }
    }
    if (ABL < 2 && s + 1 < nslab) STORE_SLAB(buf ^ 1)
    __syncthreads();
  }
  // epilogue through LDS: each wave owns a private 32 x (32+1) staging tile per N tile (reuses the operand buffers)
  float* T = smem + wave * (32 * 33);
  const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + cl] = acc[nt][r];
    // wave-private region: LDS ops of one wave complete in order; make the compiler keep the order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 32 rows x 32 cols = 256 float4: lane handles 4 float4: row = (lane>>3) + 8*i, c4 = lane&7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i, c4 = lane & 7;
      const long row = row0 + wave * 32 + rr;
      const int col = nt * 32 + c4 * 4;
      f4 v;
      v.x = T[rr * 33 + c4 * 4 + 0] + bias[col + 0]; v.y = T[rr * 33 + c4 * 4 + 1] + bias[col + 1];
      v.z = T[rr * 33 + c4 * 4 + 2] + bias[col + 2]; v.w = T[rr * 33 + c4 * 4 + 3] + bias[col + 3];
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      if (row < P && (ABL < 3 || v.x == 123.456f)) *reinterpret_cast<f4*>(C + row * ldc + col) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// v3: 256-point tile, 8 waves (one workgroup per CU): W slab staged once per 256 points (half the L2->LDS weight traffic and
// LDS stores per point), epilogue tile read back with ds_read_b128 (row stride 36 floats)
template <int NT, int ABL>
__global__ __launch_bounds__(512, 1) void gemm_v3(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                  float* __restrict__ C, long P, int K, int lda, int ldw, int ldc) {
  constexpr int BK = 16, LD = 20, BM = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][256][LD]
  float* Bs = smem + 2 * BM * LD;        // [2][NT*32][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * BM;
  const int nslab = K / BK;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f4 ra[2], rb[2];
  const int r0 = tid >> 2, c4 = (tid & 3) * 4;        // rows r0 and r0 + 128
  long ar0 = row0 + r0, ar1 = row0 + r0 + 128;
  if (ar0 >= P) ar0 = P - 1;
  if (ar1 >= P) ar1 = P - 1;
  const float* ap0 = A + ar0 * lda + c4; const float* ap1 = A + ar1 * lda + c4;
  const float* wp0 = W + (long)r0 * ldw + c4; const float* wp1 = W + (long)(r0 + 128) * ldw + c4;
#define L3(s_) { ra[0] = *reinterpret_cast<const f4*>(ap0 + (s_) * BK); ra[1] = *reinterpret_cast<const f4*>(ap1 + (s_) * BK); \
                 rb[0] = *reinterpret_cast<const f4*>(wp0 + (s_) * BK); rb[1] = *reinterpret_cast<const f4*>(wp1 + (s_) * BK); }
#define S3(b_) { *reinterpret_cast<f4*>(As + ((b_) * BM + r0) * LD + c4) = ra[0]; *reinterpret_cast<f4*>(As + ((b_) * BM + r0 + 128) * LD + c4) = ra[1]; \
                 *reinterpret_cast<f4*>(Bs + ((b_) * NT * 32 + r0) * LD + c4) = rb[0]; *reinterpret_cast<f4*>(Bs + ((b_) * NT * 32 + r0 + 128) * LD + c4) = rb[1]; }
  L3(0) S3(0) __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (ABL == 0 && s + 1 < nslab) L3(s + 1)
    const float* Ab = As + (buf * BM + wave * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* Bb = Bs + (buf * NT * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const f4 a = *reinterpret_cast<const f4*>(Ab + kb * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4*>(Bb + nt * 32 * LD + kb * 8);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
      }
    }
    if (ABL < 2 && s + 1 < nslab) S3(buf ^ 1)
    __syncthreads();
  }
  constexpr int TLD = 36;
  float* T = smem + wave * (32 * TLD);
  const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * TLD + cl] = acc[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i, cc = lane & 7;
      const long row = row0 + wave * 32 + rr;
      const int col = nt * 32 + cc * 4;
      f4 v = *reinterpret_cast<const f4*>(T + rr * TLD + cc * 4);
      const f4 bb = *reinterpret_cast<const f4*>(bias + col);
      v.x = fmaxf(v.x + bb.x, 0.f); v.y = fmaxf(v.y + bb.y, 0.f); v.z = fmaxf(v.z + bb.z, 0.f); v.w = fmaxf(v.w + bb.w, 0.f);
      if (row < P && (ABL < 3 || v.x == 123.456f)) *reinterpret_cast<f4*>(C + row * ldc + col) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}


template <int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_v4(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                     float* __restrict__ C, long P, int K, int lda, int ldw, int ldc) {
  constexpr int NT = 4, BK = 16, LD = 20;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * 128 * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * 128;
  const int col0 = blockIdx.y * 128;
  const int nslab = K / BK;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f4 ra[2], rb[2];
  const int r0 = tid >> 2, c4 = (tid & 3) * 4;
  long ar0 = row0 + r0, ar1 = row0 + r0 + 64;
  if (ar0 >= P) ar0 = P - 1;
  if (ar1 >= P) ar1 = P - 1;
  const float* ap0 = A + ar0 * lda + c4; const float* ap1 = A + ar1 * lda + c4;
  const float* wp0 = W + (long)(col0 + r0) * ldw + c4; const float* wp1 = W + (long)(col0 + r0 + 64) * ldw + c4;
#define L4(s_) { ra[0] = *reinterpret_cast<const f4*>(ap0 + (s_) * BK); ra[1] = *reinterpret_cast<const f4*>(ap1 + (s_) * BK); \
                 rb[0] = *reinterpret_cast<const f4*>(wp0 + (s_) * BK); rb[1] = *reinterpret_cast<const f4*>(wp1 + (s_) * BK); }
#define S4(b_) { *reinterpret_cast<f4*>(As + ((b_) * 128 + r0) * LD + c4) = ra[0]; *reinterpret_cast<f4*>(As + ((b_) * 128 + r0 + 64) * LD + c4) = ra[1]; \
                 *reinterpret_cast<f4*>(Bs + ((b_) * 128 + r0) * LD + c4) = rb[0]; *reinterpret_cast<f4*>(Bs + ((b_) * 128 + r0 + 64) * LD + c4) = rb[1]; }
  L4(0) S4(0) __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) L4(s + 1)
    const float* Ab = As + (buf * 128 + wave * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* Bb = Bs + (buf * 128 + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const f4 a = *reinterpret_cast<const f4*>(Ab + kb * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4*>(Bb + nt * 32 * LD + kb * 8);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
      }
    }
    if (s + 1 < nslab) S4(buf ^ 1)
    __syncthreads();
  }
  constexpr int TLD = 36;
  float* T = smem + wave * (32 * TLD);
  const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * TLD + cl] = acc[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i, cc = lane & 7;
      const long row = row0 + wave * 32 + rr;
      const int col = col0 + nt * 32 + cc * 4;
      f4 v = *reinterpret_cast<const f4*>(T + rr * TLD + cc * 4);
      const f4 bb = *reinterpret_cast<const f4*>(bias + col);
      v.x = fmaxf(v.x + bb.x, 0.f); v.y = fmaxf(v.y + bb.y, 0.f); v.z = fmaxf(v.z + bb.z, 0.f); v.w = fmaxf(v.w + bb.w, 0.f);
      if (row < P) *reinterpret_cast<f4*>(C + row * ldc + col) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}


// v5: v3 (256-point tile, 8 waves) with prefetch distance 2: loads for slab s+2 are issued while slab s is computed; two register sets
template <int NT>
__global__ __launch_bounds__(512, 1) void gemm_v5(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                  float* __restrict__ C, long P, int K, int lda, int ldw, int ldc) {
  constexpr int BK = 16, LD = 20, BM = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * BM;
  const int nslab = K / BK;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f4 xa[2], xb[2], ya[2], yb[2];
  const int r0 = tid >> 2, c4 = (tid & 3) * 4;
  long ar0 = row0 + r0, ar1 = row0 + r0 + 128;
  if (ar0 >= P) ar0 = P - 1;
  if (ar1 >= P) ar1 = P - 1;
  const float* ap0 = A + ar0 * lda + c4; const float* ap1 = A + ar1 * lda + c4;
  const float* wp0 = W + (long)r0 * ldw + c4; const float* wp1 = W + (long)(r0 + 128) * ldw + c4;
#define L5(a_, b_, s_) { a_[0] = *reinterpret_cast<const f4*>(ap0 + (s_) * BK); a_[1] = *reinterpret_cast<const f4*>(ap1 + (s_) * BK); \
                         b_[0] = *reinterpret_cast<const f4*>(wp0 + (s_) * BK); b_[1] = *reinterpret_cast<const f4*>(wp1 + (s_) * BK); }
#define S5(a_, b_, bf_) { *reinterpret_cast<f4*>(As + ((bf_) * BM + r0) * LD + c4) = a_[0]; *reinterpret_cast<f4*>(As + ((bf_) * BM + r0 + 128) * LD + c4) = a_[1]; \
                          *reinterpret_cast<f4*>(Bs + ((bf_) * NT * 32 + r0) * LD + c4) = b_[0]; *reinterpret_cast<f4*>(Bs + ((bf_) * NT * 32 + r0 + 128) * LD + c4) = b_[1]; }
#define C5(bf_)                                                                                  \
  {                                                                                              \
    const float* Ab = As + ((bf_) * BM + wave * 32 + (lane & 31)) * LD + (lane >> 5) * 4;        \
    const float* Bb = Bs + ((bf_) * NT * 32 + (lane & 31)) * LD + (lane >> 5) * 4;               \
    _Pragma("unroll") for (int kb = 0; kb < BK / 8; ++kb) {                                      \
      const f4 a = *reinterpret_cast<const f4*>(Ab + kb * 8);                                    \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                        \
        const f4 b = *reinterpret_cast<const f4*>(Bb + nt * 32 * LD + kb * 8);                   \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);              \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);              \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);              \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);              \
      }                                                                                          \
    }                                                                                            \
  }
  // x holds slab s+1 (to be published at the end of iteration s), y receives slab s+2
  L5(xa, xb, 0) S5(xa, xb, 0)
  if (nslab > 1) L5(xa, xb, 1)
  __syncthreads();
  for (int s = 0; s < nslab; s += 2) {
    if (s + 2 < nslab) L5(ya, yb, s + 2)
    C5(0)
    if (s + 1 < nslab) S5(xa, xb, 1)
    __syncthreads();
    if (s + 1 < nslab) {
      if (s + 3 < nslab) L5(xa, xb, s + 3)
      C5(1)
      if (s + 2 < nslab) S5(ya, yb, 0)
      __syncthreads();
    }
  }
  constexpr int TLD = 36;
  float* T = smem + wave * (32 * TLD);
  const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * TLD + cl] = acc[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i, cc = lane & 7;
      const long row = row0 + wave * 32 + rr;
      const int col = nt * 32 + cc * 4;
      f4 v = *reinterpret_cast<const f4*>(T + rr * TLD + cc * 4);
      const f4 bb = *reinterpret_cast<const f4*>(bias + col);
      v.x = fmaxf(v.x + bb.x, 0.f); v.y = fmaxf(v.y + bb.y, 0.f); v.z = fmaxf(v.z + bb.z, 0.f); v.w = fmaxf(v.w + bb.w, 0.f);
      if (row < P) *reinterpret_cast<f4*>(C + row * ldc + col) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <class KFn>
static double time_kernel(KFn fn, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) fn();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const long P = argc > 1 ? atol(argv[1]) : 524288;
  const int K = 256, N = 256;
  std::vector<float> hA((size_t)P * K), hW((size_t)N * K), hb(N);
  srand(1);
  for (auto& x : hA) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : hW) x = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.1f;
  for (auto& x : hb) x = (rand() / (float)RAND_MAX) * 0.1f;
  float *A, *W, *b, *C;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&C, (size_t)P * N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
  const unsigned grid = (unsigned)((P + 127) / 128);
  const double flop = 2.0 * P * N * K;
  auto check = [&](const char* name) {
    std::vector<float> hC(256 * (size_t)N);
    CK(hipMemcpy(hC.data(), C + (size_t)(P - 256) * N, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int r = 0; r < 256; r += 37) for (int n = 0; n < N; n += 11) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(P - 256 + r) * K + k] * hW[(size_t)n * K + k];
      if (s < 0) s = 0;
      maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n]));
    }
    printf("  %-28s check maxerr %.2e\n", name, maxerr);
  };
#define RUN(NAME, KERNEL, BKV)                                                                                    \
  {                                                                                                               \
    size_t lds = (size_t)(2 * 128 * (BKV + 4) + 2 * 256 * (BKV + 4)) * 4;                                         \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    CK(hipMemset(C, 0, (size_t)P * N * 4));                                                                       \
    double ms = time_kernel([&] { hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), lds, 0, A, W, b, C, P, K, K, K, N); }, 10); \
    CK(hipGetLastError());                                                                                        \
    printf("%-28s %.3f ms  %.1f TF/s  (lds %zu)\n", NAME, ms, flop / (ms * 1e-3) / 1e12, lds);                    \
    check(NAME);                                                                                                  \
  }
  RUN("v1 BK16 2w/simd", (gemm_v1<8, 16, 2>), 16)
  RUN("v1 BK32 1w/simd", (gemm_v1<8, 32, 1>), 32)
  RUN("v1 BK16 1w/simd", (gemm_v1<8, 16, 1>), 16)
  RUN("v2 BK16 2w/simd (LDS epi)", (gemm_v2<8, 16, 2, 0>), 16)
  RUN("v2 abl1 no global loads", (gemm_v2<8, 16, 2, 1>), 16)
  RUN("v2 abl2 no loads/LDS stores", (gemm_v2<8, 16, 2, 2>), 16)
  RUN("v2 abl3 + no C stores", (gemm_v2<8, 16, 2, 3>), 16)

#define RUN3(NAME, KERNEL)                                                                                        \
  {                                                                                                               \
    size_t lds = (size_t)(2 * 256 * 20 + 2 * 256 * 20) * 4;                                                       \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    CK(hipMemset(C, 0, (size_t)P * N * 4));                                                                       \
    double ms = time_kernel([&] { hipLaunchKernelGGL(KERNEL, dim3((unsigned)((P + 255) / 256)), dim3(512), lds, 0, A, W, b, C, P, K, K, K, N); }, 10); \
    CK(hipGetLastError());                                                                                        \
    printf("%-28s %.3f ms  %.1f TF/s  (lds %zu)\n", NAME, ms, flop / (ms * 1e-3) / 1e12, lds);                    \
    check(NAME);                                                                                                  \
  }
  RUN3("v3 BM256 8 waves", (gemm_v3<8, 0>))
  RUN3("v3 abl1 no loads", (gemm_v3<8, 1>))
  RUN3("v3 abl3 no loads/stores", (gemm_v3<8, 3>))

#define RUN4(NAME, KERNEL)                                                                                        \
  {                                                                                                               \
    size_t lds = (size_t)(2 * 128 * 20 + 2 * 128 * 20) * 4;                                                       \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    CK(hipMemset(C, 0, (size_t)P * N * 4));                                                                       \
    double ms = time_kernel([&] { hipLaunchKernelGGL(KERNEL, dim3((unsigned)((P + 127) / 128), 2), dim3(256), lds, 0, A, W, b, C, P, K, K, K, N); }, 10); \
    CK(hipGetLastError());                                                                                        \
    printf("%-28s %.3f ms  %.1f TF/s  (lds %zu)\n", NAME, ms, flop / (ms * 1e-3) / 1e12, lds);                    \
    check(NAME);                                                                                                  \
  }
  RUN3("v5 BM256 prefetch dist 2", (gemm_v5<8>))
  RUN4("v4 128x128 tiles 3wg/CU", (gemm_v4<3>))
  RUN4("v4 128x128 tiles 4wg/CU", (gemm_v4<4>))
  return 0;
}
