// Does a second staging register set (activation tiles fetched two ahead) help the weight-stationary f16x3 layer GEMM?
// Plain DIRECT -> bias+ReLU layer, 524288 x 256 x 256.  Build: hipcc --offload-arch=gfx950 -O3 ws_depth_probe.hip -o ws_depth_probe
#include <hip/hip_runtime.h>
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
  float* RSR = reinterpret_cast<float*>(smem + 2 * ABUF + 8 * 32 * 36 * 4);   // ring of 4 x 32 row scales (software-pipelined epilogue)
  int rs_slot = 0;
#define LOADT(R_, t_) { long row = ((t_) * TP) + srow; if (row >= P) row = P - 1; const float* ap = A + row * 256 + sc4 * 4; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) R_[i] = *reinterpret_cast<const f4*>(ap + i * 64); }
#define STORET(R_, buf_) { float mx = 0.f; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) mx = fmaxf(fmaxf(fmaxf(fabsf(R_[i].x), fabsf(R_[i].y)), fmaxf(fabsf(R_[i].z), fabsf(R_[i].w))), mx); \
    _Pragma("unroll") for (int d = 8; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 16)); \
    const float sc = pow2_scale_for(mx); unsigned char* base = smem + (buf_) * ABUF + srow * ALD + sc4 * 8; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { f16x4 h1, h2; float x; \
      x = R_[i].x * sc; h1[0] = (_Float16)x; h2[0] = (_Float16)(x - (float)h1[0]); x = R_[i].y * sc; h1[1] = (_Float16)x; h2[1] = (_Float16)(x - (float)h1[1]); \
      x = R_[i].z * sc; h1[2] = (_Float16)x; h2[2] = (_Float16)(x - (float)h1[2]); x = R_[i].w * sc; h1[3] = (_Float16)x; h2[3] = (_Float16)(x - (float)h1[3]); \
      *reinterpret_cast<f16x4*>(base + i * 128) = h1; *reinterpret_cast<f16x4*>(base + APLANE + i * 128) = h2; } \
    if (sc4 == 0) { reinterpret_cast<float*>(smem + (buf_) * ABUF + 2 * APLANE)[srow] = 1.0f / sc; RSR[rs_slot * 32 + srow] = 1.0f / sc; } }
#define COMPUTE(t_, buf_) { f32x16 acc; _Pragma("unroll") for (int j = 0; j < 16; ++j) acc[j] = 0.f; \
    const unsigned char* Ab = smem + (buf_) * ABUF + (lane & 31) * ALD + (lane >> 5) * 16; \
    _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) { const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32); const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + APLANE + kb * 32); \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0); } \
    const float* rs = reinterpret_cast<const float*>(smem + (buf_) * ABUF + 2 * APLANE); const int hi = lane >> 5, cl = lane & 31; \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + cl] = acc[r]; \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { const int rr = (lane >> 3) + 8 * i, cc = lane & 7; const long row = (t_) * TP + rr; const float rsc = rs[rr]; \
      f4 v = *reinterpret_cast<const f4*>(T + rr * 36 + cc * 4); \
      v.x = fmaxf(v.x * (rsc * ws4.x) + bias4.x, 0.f); v.y = fmaxf(v.y * (rsc * ws4.y) + bias4.y, 0.f); v.z = fmaxf(v.z * (rsc * ws4.z) + bias4.z, 0.f); v.w = fmaxf(v.w * (rsc * ws4.w) + bias4.w, 0.f); \
      *reinterpret_cast<f4*>(C + (row < P ? row : P - 1) * 256 + wave * 32 + cc * 4) = v; } \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
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
  } else if (DEPTH == 4) {
    // DEPTH 3 + the epilogue of tile t - 1 cut into 16 slices that sit BETWEEN the k16 blocks of tile t's MFMAs (scheduling fences pin
    // the order), so that a wave's VALU / LDS / store instructions issue while its own MFMAs execute.
    const long tl = t1 - 1;
    f32x16 acc;
    float* T = reinterpret_cast<float*>(smem + 2 * ABUF) + wave * (32 * 36);
    const int hi = lane >> 5, cl = lane & 31, er = lane >> 3, ec = lane & 7;
#define MF_ONLY(buf_) { _Pragma("unroll") for (int j = 0; j < 16; ++j) acc[j] = 0.f; \
      const unsigned char* Ab = smem + (buf_) * ABUF + (lane & 31) * ALD + (lane >> 5) * 16; \
      _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) { const f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab + kb * 32); const f16x8 a2 = *reinterpret_cast<const f16x8*>(Ab + APLANE + kb * 32); \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0); } }
#define T_WRITE() { _Pragma("unroll") for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + cl] = acc[r]; \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#define EPI_ONLY(tp_, slot_) { const float* rs = RSR + (slot_) * 32; \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { const int rr = er + 8 * i; const float rsc = rs[rr]; f4 v = *reinterpret_cast<const f4*>(T + rr * 36 + ec * 4); \
        v.x = fmaxf(v.x * (rsc * ws4.x) + bias4.x, 0.f); v.y = fmaxf(v.y * (rsc * ws4.y) + bias4.y, 0.f); v.z = fmaxf(v.z * (rsc * ws4.z) + bias4.z, 0.f); v.w = fmaxf(v.w * (rsc * ws4.w) + bias4.w, 0.f); \
        *reinterpret_cast<f4*>(C + ((tp_) * TP + rr) * 256 + wave * 32 + ec * 4) = v; } \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    // MFMAs of the tile in buffer buf_ interleaved with the epilogue of tile tp_ (its accumulators already sit in T)
#define MF_EPI(buf_, tp_, slot_) { _Pragma("unroll") for (int j = 0; j < 16; ++j) acc[j] = 0.f; \
      const unsigned char* Ab = smem + (buf_) * ABUF + (lane & 31) * ALD + (lane >> 5) * 16; \
      const float* rs = RSR + (slot_) * 32; \
      f4 ev[4]; float ersc[4]; \
      f16x8 a1 = *reinterpret_cast<const f16x8*>(Ab), a2 = *reinterpret_cast<const f16x8*>(Ab + APLANE); \
      _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) { \
        f16x8 n1 = a1, n2 = a2; \
        if (kb + 1 < KB) { n1 = *reinterpret_cast<const f16x8*>(Ab + (kb + 1) * 32); n2 = *reinterpret_cast<const f16x8*>(Ab + APLANE + (kb + 1) * 32); } \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w2[kb], acc, 0, 0, 0); \
        { const int i = kb >> 2, q = kb & 3; const int rr = er + 8 * i; \
          if (q == 0) { ersc[i] = rs[rr]; ev[i] = *reinterpret_cast<const f4*>(T + rr * 36 + ec * 4); } \
          if (q == 1) { ev[i].x = fmaxf(ev[i].x * (ersc[i] * ws4.x) + bias4.x, 0.f); ev[i].y = fmaxf(ev[i].y * (ersc[i] * ws4.y) + bias4.y, 0.f); } } \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, w1[kb], acc, 0, 0, 0); \
        { const int i = kb >> 2, q = kb & 3; \
          if (q == 2) { ev[i].z = fmaxf(ev[i].z * (ersc[i] * ws4.z) + bias4.z, 0.f); ev[i].w = fmaxf(ev[i].w * (ersc[i] * ws4.w) + bias4.w, 0.f); } } \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w1[kb], acc, 0, 0, 0); \
        { const int i = kb >> 2, q = kb & 3; const int rr = er + 8 * i; \
          if (q == 3) *reinterpret_cast<f4*>(C + ((tp_) * TP + rr) * 256 + wave * 32 + ec * 4) = ev[i]; } \
        a1 = n1; a2 = n2; \
        __builtin_amdgcn_sched_barrier(0); \
      } \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#define CL(t_) ((t_) < tl ? (t_) : tl)
#define SLOT(t_) ((int)(((t_) - tile0) & 3))
    // (the preamble put tile0 into buffer 0 with rs_slot 0)
    LOADT(rb, CL(tile0 + 1))
    if (late) LOADT(ra, CL(tile0 + 2))
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
    if (!late) {
      // first tile: MFMAs only
      LOADT(ra, CL(tile0 + 2)) MF_ONLY(0) rs_slot = 1; STORET(rb, 1)
      __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
      for (long t = tile0 + 1; t < t1; t += 2) {
        T_WRITE() LOADT(rb, CL(t + 2)) MF_EPI(1, t - 1, SLOT(t - 1)) rs_slot = SLOT(t + 1); STORET(ra, 0)
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
        T_WRITE() LOADT(ra, CL(t + 3)) MF_EPI(0, t, SLOT(t)) rs_slot = SLOT(t + 2); STORET(rb, 1)
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
      }
    } else {
      rs_slot = 1; STORET(rb, 1) LOADT(rb, CL(tile0 + 3)) MF_ONLY(0)
      __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
      for (long t = tile0 + 1; t < t1; t += 2) {
        rs_slot = SLOT(t + 1); STORET(ra, 0) LOADT(ra, CL(t + 3)) T_WRITE() MF_EPI(1, t - 1, SLOT(t - 1))
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
        rs_slot = SLOT(t + 2); STORET(rb, 1) LOADT(rb, CL(t + 4)) T_WRITE() MF_EPI(0, t, SLOT(t))
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
      }
    }
    // tpw is even: tiles tile0 .. t1 - 2 have been written; the last tile's accumulators are in acc (tile t1 - 1 was computed as the
    // second half of the last trip: its epilogue is still due)
    T_WRITE() EPI_ONLY(t1 - 1, SLOT(t1 - 1))
  } else if (DEPTH == 3) {
    // as DEPTH 2, but one loop per wave group and unconditional prefetches (tile index clamped to the last tile): every path through a
    // loop issues the same memory operations in the same order, so the compiler can tell how many younger operations may stay in flight
    // when a staging set is consumed (otherwise it falls back to s_waitcnt vmcnt(0): the epilogue stores drain once per tile)
    const long tl = t1 - 1;
#define CL(t_) ((t_) < tl ? (t_) : tl)
    LOADT(rb, CL(tile0 + 1))
    if (late) LOADT(ra, CL(tile0 + 2))
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
    if (!late) {
      for (long t = tile0; t < t1; t += 2) {
        LOADT(ra, CL(t + 2))
        COMPUTE(t, 0)
        STORET(rb, 1)
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
        LOADT(rb, CL(t + 3))
        COMPUTE(t + 1, 1)
        STORET(ra, 0)
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
      }
    } else {
#define LATE_BODY(t) \
        STORET(rb, 1) \
        LOADT(rb, CL((t) + 3)) \
        COMPUTE((t), 0) \
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); \
        STORET(ra, 0) \
        LOADT(ra, CL((t) + 4)) \
        COMPUTE((t) + 1, 1) \
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
      LATE_BODY(tile0)   // peeled: the loop header then only sees the steady state
      for (long t = tile0 + 2; t < t1; t += 2) { LATE_BODY(t) }
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
  const size_t lds = (size_t)2 * ABUF + 8 * 32 * 36 * 4 + 4 * 32 * 4;
  for (int depth = 2; depth <= 4; ++depth) {
    auto kern = depth == 2 ? ws_gemm<2> : depth == 3 ? ws_gemm<3> : ws_gemm<4>;
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
    printf("prefetch depth %d: %.3f ms  %.0f GB/s  maxerr %.2e\n", depth, ms, 2.0 * P * 1024 / (ms * 1e-3) / 1e9, maxerr);
  }
  return 0;
}
