// Probe 2: bf16x6 GEMM with (a) activations loaded straight into MFMA operand registers (each wave owns its 32 rows: no LDS,
// no sharing needed), (b) pre-split weight planes through double-buffered LDS, (c) prefetch distance of two k16 stages.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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
// 8 floats (two f4) -> three bf16x8 operand fragments
__device__ __forceinline__ void split8(const f4& lo, const f4& hi, bf16x8& o1, bf16x8& o2, bf16x8& o3) {
  unsigned int a[8], b[8], c[8];
  split3(lo.x, a[0], b[0], c[0]); split3(lo.y, a[1], b[1], c[1]); split3(lo.z, a[2], b[2], c[2]); split3(lo.w, a[3], b[3], c[3]);
  split3(hi.x, a[4], b[4], c[4]); split3(hi.y, a[5], b[5], c[5]); split3(hi.z, a[6], b[6], c[6]); split3(hi.w, a[7], b[7], c[7]);
  u32x4 p1 = {a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16)};
  u32x4 p2 = {b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16)};
  u32x4 p3 = {c[0] | (c[1] << 16), c[2] | (c[3] << 16), c[4] | (c[5] << 16), c[6] | (c[7] << 16)};
  o1 = __builtin_bit_cast(bf16x8, p1); o2 = __builtin_bit_cast(bf16x8, p2); o3 = __builtin_bit_cast(bf16x8, p3);
}

template <int NT, int ABL>
__global__ __launch_bounds__(256, 2) void gemm_bf16x6_v2(const float* __restrict__ A, const unsigned short* __restrict__ W1,
                                                          const unsigned short* __restrict__ W2, const unsigned short* __restrict__ W3,
                                                          const float* __restrict__ bias, float* __restrict__ C, long P, int K, int N) {
  constexpr int LDB = 48;                       // bytes per LDS row: 16 bf16 + 16 pad
  constexpr int BBUF = 3 * NT * 32 * LDB;       // one stage of the three weight planes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)blockIdx.x * 128;
  const int nstage = K / 16;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  // A: lane (row = lane&31, half = lane>>5) reads 8 consecutive floats per stage
  long arow = row0 + wave * 32 + (lane & 31);
  if (arow >= P) arow = P - 1;
  const float* ap = A + arow * K + (lane >> 5) * 8;
  // W planes: one stage = NT*32 rows x 32 B per plane = NT*64 16-byte pieces per plane; 256 threads -> NT/4 pieces per plane per thread
  constexpr int NW = NT * 64 / 256;             // = 2 for NT = 8
  const int wr = tid >> 1, wc = tid & 1;        // piece i of a plane: row = wr + i*128, 16-byte column wc
  f4 a0lo, a0hi, a1lo, a1hi;                    // raw A of the two stages in flight
  u32x4 b0[3 * NW], b1[3 * NW];                 // raw W pieces of the two stages in flight
#define LOAD_A(lo_, hi_, st_) { lo_ = *reinterpret_cast<const f4*>(ap + (st_) * 16); hi_ = *reinterpret_cast<const f4*>(ap + (st_) * 16 + 4); }
#define LOAD_B(bb_, st_)                                                                              \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                  \
      const long off = (long)(wr + i * 128) * K + (st_) * 16 + wc * 8;                                \
      bb_[i] = *reinterpret_cast<const u32x4*>(W1 + off);                                             \
      bb_[NW + i] = *reinterpret_cast<const u32x4*>(W2 + off);                                        \
      bb_[2 * NW + i] = *reinterpret_cast<const u32x4*>(W3 + off);                                    \
    }                                                                                                 \
  }
#define STORE_B(bb_, buf_)                                                                            \
  {                                                                                                   \
    unsigned char* Bs = smem + (buf_) * BBUF;                                                         \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                  \
      const int r = wr + i * 128;                                                                     \
      *reinterpret_cast<u32x4*>(Bs + (0 * NT * 32 + r) * LDB + wc * 16) = bb_[i];                     \
      *reinterpret_cast<u32x4*>(Bs + (1 * NT * 32 + r) * LDB + wc * 16) = bb_[NW + i];                \
      *reinterpret_cast<u32x4*>(Bs + (2 * NT * 32 + r) * LDB + wc * 16) = bb_[2 * NW + i];            \
    }                                                                                                 \
  }
#define COMPUTE(a1_, a2_, a3_, buf_)                                                                  \
  {                                                                                                   \
    const unsigned char* Bb = smem + (buf_) * BBUF + (lane & 31) * LDB + (lane >> 5) * 16;            \
    _Pragma("unroll") for (int np = 0; np < NT; np += 2) {                                            \
      const bf16x8 u1 = *reinterpret_cast<const bf16x8*>(Bb + (0 * NT * 32 + np * 32) * LDB);         \
      const bf16x8 u2 = *reinterpret_cast<const bf16x8*>(Bb + (1 * NT * 32 + np * 32) * LDB);         \
      const bf16x8 u3 = *reinterpret_cast<const bf16x8*>(Bb + (2 * NT * 32 + np * 32) * LDB);         \
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(Bb + (0 * NT * 32 + np * 32 + 32) * LDB);    \
      const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(Bb + (1 * NT * 32 + np * 32 + 32) * LDB);    \
      const bf16x8 v3 = *reinterpret_cast<const bf16x8*>(Bb + (2 * NT * 32 + np * 32 + 32) * LDB);    \
      acc[np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, u3, acc[np], 0, 0, 0);                   \
      acc[np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, v3, acc[np + 1], 0, 0, 0);           \
      acc[np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_, u1, acc[np], 0, 0, 0);                   \
      acc[np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_, v1, acc[np + 1], 0, 0, 0);           \
      acc[np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_, u2, acc[np], 0, 0, 0);                   \
      acc[np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_, v2, acc[np + 1], 0, 0, 0);           \
      acc[np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, u2, acc[np], 0, 0, 0);                   \
      acc[np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, v2, acc[np + 1], 0, 0, 0);           \
      acc[np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_, u1, acc[np], 0, 0, 0);                   \
      acc[np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_, v1, acc[np + 1], 0, 0, 0);           \
      acc[np] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, u1, acc[np], 0, 0, 0);                   \
      acc[np + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, v1, acc[np + 1], 0, 0, 0);           \
    }                                                                                                 \
  }
  // prologue: stages 0 and 1 in flight, stage 0 of W into LDS
  LOAD_A(a0lo, a0hi, 0) LOAD_B(b0, 0)
  if (nstage > 1) { LOAD_A(a1lo, a1hi, 1) LOAD_B(b1, 1) }
  STORE_B(b0, 0)
  __syncthreads();
  for (int s = 0; s < nstage; s += 2) {
    {   // even stage s: A from set 0, W from LDS buf 0; prefetch stage s+2 into set 0; publish stage s+1 (set 1) to LDS buf 1
      bf16x8 x1, x2, x3;
      split8(a0lo, a0hi, x1, x2, x3);
      if (s + 2 < nstage) { if (ABL != 1 && ABL < 4) LOAD_A(a0lo, a0hi, s + 2) if (ABL < 2) LOAD_B(b0, s + 2) }
      COMPUTE(x1, x2, x3, 0)
      if (ABL < 3 && s + 1 < nstage) STORE_B(b1, 1)
      __syncthreads();
    }
    if (s + 1 < nstage) {   // odd stage s+1
      bf16x8 x1, x2, x3;
      split8(a1lo, a1hi, x1, x2, x3);
      if (s + 3 < nstage) { if (ABL != 1 && ABL < 4) LOAD_A(a1lo, a1hi, s + 3) if (ABL < 2) LOAD_B(b1, s + 3) }
      COMPUTE(x1, x2, x3, 1)
      if (ABL < 3 && s + 2 < nstage) STORE_B(b0, 0)
      __syncthreads();
    }
  }
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
      if (row < P && (ABL < 5 || v.x == 123.456f)) *reinterpret_cast<f4*>(C + row * N + col) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int DUMMY>
__global__ __launch_bounds__(256, 2) void gemm_bf16x6_v3(const float* __restrict__ A, const unsigned short* __restrict__ W1,
                                                          const unsigned short* __restrict__ W2, const unsigned short* __restrict__ W3,
                                                          const float* __restrict__ bias, float* __restrict__ C, long P, int K, int N) {
  constexpr int NT = 8, LDB = 48, BBUF = 3 * NT * 32 * LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;      // wave tile: rows [wm*64, +64), cols [wn*128, +128)
  const long row0 = (long)blockIdx.x * 128;
  const int nstage = K / 16;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  long ar0 = row0 + wm * 64 + (lane & 31), ar1 = ar0 + 32;
  if (ar0 >= P) ar0 = P - 1;
  if (ar1 >= P) ar1 = P - 1;
  const float* ap0 = A + ar0 * K + (lane >> 5) * 8;
  const float* ap1 = A + ar1 * K + (lane >> 5) * 8;
  constexpr int NW = NT * 64 / 256;
  const int wr = tid >> 1, wc = tid & 1;
  f4 a0[4], a1[4];                              // raw A (2 row tiles x lo/hi) of the two stages in flight
  u32x4 b0[3 * NW], b1[3 * NW];
#define LOAD_A3(aa_, st_) { aa_[0] = *reinterpret_cast<const f4*>(ap0 + (st_) * 16); aa_[1] = *reinterpret_cast<const f4*>(ap0 + (st_) * 16 + 4); \
                            aa_[2] = *reinterpret_cast<const f4*>(ap1 + (st_) * 16); aa_[3] = *reinterpret_cast<const f4*>(ap1 + (st_) * 16 + 4); }
#define COMPUTE3(xa_, xb_, buf_)                                                                      \
  {                                                                                                   \
    const unsigned char* Bb = smem + (buf_) * BBUF + (wn * 128 + (lane & 31)) * LDB + (lane >> 5) * 16; \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                \
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(Bb + (0 * NT * 32 + nt * 32) * LDB);         \
      const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(Bb + (1 * NT * 32 + nt * 32) * LDB);         \
      const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(Bb + (2 * NT * 32 + nt * 32) * LDB);         \
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa_[0], w3, acc[0][nt], 0, 0, 0);          \
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb_[0], w3, acc[1][nt], 0, 0, 0);          \
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa_[2], w1, acc[0][nt], 0, 0, 0);          \
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb_[2], w1, acc[1][nt], 0, 0, 0);          \
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa_[1], w2, acc[0][nt], 0, 0, 0);          \
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb_[1], w2, acc[1][nt], 0, 0, 0);          \
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa_[0], w2, acc[0][nt], 0, 0, 0);          \
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb_[0], w2, acc[1][nt], 0, 0, 0);          \
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa_[1], w1, acc[0][nt], 0, 0, 0);          \
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb_[1], w1, acc[1][nt], 0, 0, 0);          \
      acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa_[0], w1, acc[0][nt], 0, 0, 0);          \
      acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb_[0], w1, acc[1][nt], 0, 0, 0);          \
    }                                                                                                 \
  }
  LOAD_A3(a0, 0) LOAD_B(b0, 0)
  if (nstage > 1) { LOAD_A3(a1, 1) LOAD_B(b1, 1) }
  STORE_B(b0, 0)
  __syncthreads();
  for (int s = 0; s < nstage; s += 2) {
    {
      bf16x8 xa[3], xb[3];
      split8(a0[0], a0[1], xa[0], xa[1], xa[2]);
      split8(a0[2], a0[3], xb[0], xb[1], xb[2]);
      if (s + 2 < nstage) { LOAD_A3(a0, s + 2) LOAD_B(b0, s + 2) }
      COMPUTE3(xa, xb, 0)
      if (s + 1 < nstage) STORE_B(b1, 1)
      __syncthreads();
    }
    if (s + 1 < nstage) {
      bf16x8 xa[3], xb[3];
      split8(a1[0], a1[1], xa[0], xa[1], xa[2]);
      split8(a1[2], a1[3], xb[0], xb[1], xb[2]);
      if (s + 3 < nstage) { LOAD_A3(a1, s + 3) LOAD_B(b1, s + 3) }
      COMPUTE3(xa, xb, 1)
      if (s + 2 < nstage) STORE_B(b0, 0)
      __syncthreads();
    }
  }
  float* T = reinterpret_cast<float*>(smem) + wave * (32 * 33);
  const int hi = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + cl] = acc[mt][nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i, c4 = lane & 7;
      const long row = row0 + wm * 64 + mt * 32 + rr;
      const int col = wn * 128 + nt * 32 + c4 * 4;
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
  size_t lds = (size_t)2 * 3 * 256 * 48;
  for (int variant = 0; variant < 6; ++variant) {
  auto kernel = variant == 0 ? gemm_bf16x6_v2<8, 0> : variant == 1 ? gemm_bf16x6_v2<8, 1> : variant == 2 ? gemm_bf16x6_v2<8, 2> : variant == 3 ? gemm_bf16x6_v2<8, 3> : variant == 4 ? gemm_bf16x6_v2<8, 4> : gemm_bf16x6_v2<8, 5>;
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
  double maxerr = 0;
  for (int r = 0; r < 256; r += 5) for (int n = 0; n < N; n += 3) {
    double s = hb[n];
    for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(P - 256 + r) * K + k] * hW[(size_t)n * K + k];
    if (s < 0) s = 0;
    maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n]));
  }
  printf("bf16x6 variant %d: %.3f ms  %.1f TF/s-equivalent  maxerr %.2e  lds %zu\n", variant, ms, flop / (ms * 1e-3) / 1e12, maxerr, lds);
  }
  return 0;
}
