// shared error bookkeeping of the HIP translation units
#pragma once
#include <hip/hip_runtime.h>
namespace cnr {
extern hipError_t g_first_error;
extern const char* g_first_error_where;
// optional per-launch timing with HIP events on the launch stream (bench / profiling only; off by default)
void timing_begin(const char* name, int kind, int nt, long P, int N, int K, int pairs, hipStream_t s, double bytes);
void timing_end(hipStream_t s);
// hipFuncSetAttribute is a per-device setting: remember per (kernel instantiation, device) that it has been applied.
// Usage: static DeviceOnce once; if (once.first()) hipFuncSetAttribute(...);
struct DeviceOnce {
  bool done[64] = {};
  bool first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;   // unknown device: just apply the attribute again
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};
struct TimingScope {
  hipStream_t s;
  TimingScope(const char* name, int kind, int nt, long P, int N, int K, int pairs, hipStream_t st, double bytes = 0.0) : s(st) {
    timing_begin(name, kind, nt, P, N, K, pairs, s, bytes);
  }
  ~TimingScope() { timing_end(s); }
};
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global access of the wave
// (s_waitcnt vmcnt(0) in front of s_barrier): in the streaming GEMM kernels that makes the tile prefetched for a later iteration
// land before EVERY barrier, i.e. the prefetch distance can never exceed one iteration.  Those kernels exchange data between waves
// through LDS only (global data written by a wave is never read by another wave of the same launch), so no global ordering is needed.
__device__ __forceinline__ void cnr_lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// 1 / s for a power of two s (every row / operand scale of the split-f16 arithmetic is one, 2^-114 .. 2^114): exponent arithmetic, exact -- the IEEE
// division the compiler emits for `1.0f / s` is ~10 VALU instructions (v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) per row and tile
__device__ __forceinline__ float cnr_pow2_rcp(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }
// max over the 16 lanes of a DPP row (every lane gets it) by four data-parallel-primitive moves (quad_perm xor 1, xor 2, row_half_mirror, row_mirror)
// instead of four __shfl_xor(.., 16), which hipcc lowers to ds_bpermute_b32: an LDS-crossbar round trip each, on the path between a tile's
// arrival and its f16 planes.  max is associative and commutative: bit-identical to the shuffle form.
__device__ __forceinline__ float cnr_max16(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  int v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false))); v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false))); v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false))); v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false)));
#endif
  return x;
}
// Cross-half and cross-row exchanges as VALU ops (gfx950: v_permlane32_swap_b32 / v_permlane16_swap_b32) instead of __shfl_xor(.., 32) / (.., 16), which hipcc
// lowers to ds_bpermute_b32 (an LDS-crossbar round trip each, and an s_waitcnt on the LDS counter in the middle of an epilogue).  permlane32_swap(x, x) returns
// {(lo, lo), (hi, hi)} (lanes 0-31 | 32-63), permlane16_swap(x, x) {(r0, r0, r2, r2), (r1, r1, r3, r3)} (16-lane rows): the two results hold this lane's value and
// its partner's, so max / min / sum of the pair are one more VALU op -- the same numbers as with the shuffle (max, min exact; a + b commutative).
__device__ __forceinline__ float cnr_pair32_max(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#endif
  return x;
}
__device__ __forceinline__ float cnr_pair32_sum(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
  return x;
}
__device__ __forceinline__ int cnr_pair32_min(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
  const int a = (int)r[0], b = (int)r[1];
  x = a < b ? a : b;
#endif
  return x;
}
__device__ __forceinline__ int cnr_pair16_min(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
  const int a = (int)r[0], b = (int)r[1];
  x = a < b ? a : b;
#endif
  return x;
}
// max over the 8 lanes of a half DPP row / min over the 16 lanes of a DPP row by DPP moves (see cnr_max16), and the pair exchanges of two values at once
__device__ __forceinline__ float cnr_max8(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  int v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false))); v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false))); v = __float_as_int(x);
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false)));
#endif
  return x;
}
__device__ __forceinline__ int cnr_min16(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  int o;
  o = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false); x = o < x ? o : x;
  o = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false); x = o < x ? o : x;
  o = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false); x = o < x ? o : x;
  o = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false); x = o < x ? o : x;
#endif
  return x;
}
__device__ __forceinline__ int cnr_ror8_min(int x) {   // min with the lane 8 further in the 16-lane row (lane ^ 8)
#if defined(__HIP_DEVICE_COMPILE__)
  const int o = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);
  x = o < x ? o : x;
#endif
  return x;
}
#define CNR_LAUNCH_CHECK(where)                                   \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess && cnr::g_first_error == hipSuccess) {   \
      cnr::g_first_error = e_;                                    \
      cnr::g_first_error_where = where;                           \
    }                                                             \
  } while (0)
