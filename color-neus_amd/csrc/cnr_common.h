// Common definitions for the Color-NeuS MI355X renderer library.
//
// The same sources build twice:
//   * hipcc --offload-arch=gfx950          -> libcolorneus_hip.so   (THE product; all arithmetic in HIP kernels)
//   * g++ -DCNR_CPU_EMU                    -> libcolorneus_emu.so   (test-only: runs the identical host orchestration,
//                                             operand views, epilogues and per-point bodies on the CPU so that the
//                                             host logic is testable without a GPU.  Never loaded by the product path.)
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include "cnr_debug.h"

#if defined(CNR_CPU_EMU)
#define CNR_HD inline
#define CNR_D inline
typedef void* cnr_stream;
#else
#include <hip/hip_runtime.h>
#define CNR_HD __host__ __device__ __forceinline__
#define CNR_D __device__ __forceinline__
typedef hipStream_t cnr_stream;
#endif

namespace cnr {

#if defined(CNR_CPU_EMU)
struct alignas(16) f4 {
  float x, y, z, w;
};
#else
// native 4-wide vector: arrays of this type stay in VGPRs (arrays of an aligned struct were placed in scratch by hipcc)
typedef float f4 __attribute__((ext_vector_type(4)));
#endif

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr int kMaxLayers = 12;   // per MLP
constexpr int kEmb = 48;         // padded width of the SDF positional-encoding buffer (39 -> 48: GEMM operands are padded to 16)
constexpr int kAux = 48;         // padded width of the auxiliary input buffer [p(3) g(3) PE4(dir)(27) pad]
constexpr int kTop = 16;         // padded width of the 3-wide cotangent buffers that feed GEMMs

CNR_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

#if defined(CNR_CPU_EMU)
CNR_HD float fast_exp_(float x) { return expf(x); }
CNR_HD float fast_log_(float x) { return logf(x); }
CNR_HD float fast_rcp_(float x) { return 1.0f / x; }
#else
// hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp): used only inside the GEMM prologues /
// epilogues on softplus(beta=100) terms, where the affected quantity is bounded (log term <= 0.0069, sigmoid in (0,1))
// so the absolute error stays below 1e-9 / 1e-7 -- far inside the fp32 round-off of the surrounding dot products
CNR_HD float fast_exp_(float x) { return __expf(x); }
CNR_HD float fast_log_(float x) { return __logf(x); }
CNR_HD float fast_rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

// nn.Softplus(beta=100, threshold=20)                                      (reference fields.py:77)
// log1p(exp(t))/100 == max(z,0) + log(1 + exp(-|t|))/100 for every t; above the threshold the reference returns z
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CNR_CPU_EMU)
// Device versions: straight-line code on the raw v_exp_f32 (2^x) / v_log_f32 (log2) / v_rcp_f32 units -- no range-reduction or
// denormal fix-ups (the arguments are in (0,1] and [1,2]) and no branches: above the threshold exp(-t) < 2^-24, so 1 + e rounds to 1
// and the log term / the sigmoid complement vanish exactly, which is the reference's threshold behaviour.
CNR_HD float softplus100(float z) {
  const float e = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);            // exp(-|100 z|)
  return fmaxf(z, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * 0.0069314718055994531f;  // log2(1 + e) * ln2 / 100
}
// d softplus / dz = sigmoid(100 z) (1 above the threshold)
CNR_HD float softplus100_d1(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -144.26950408889634f));
}
// d2 softplus / dz2 = 100 s (1-s) (0 above the threshold)
CNR_HD float softplus100_d2(float z) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -144.26950408889634f));
  return 100.0f * s * (1.0f - s);
}
#else
CNR_HD float softplus100(float z) {
  float t = 100.0f * z;
  if (t > 20.0f) return z;
  return fmaxf(z, 0.0f) + fast_log_(1.0f + fast_exp_(-fabsf(t))) * 0.01f;
}
// d softplus / dz = sigmoid(100 z) (1 above the threshold)
CNR_HD float softplus100_d1(float z) {
  float t = 100.0f * z;
  return t > 20.0f ? 1.0f : fast_rcp_(1.0f + fast_exp_(-t));
}
// d2 softplus / dz2 = 100 s (1-s) (0 above the threshold)
CNR_HD float softplus100_d2(float z) {
  float t = 100.0f * z;
  if (t > 20.0f) return 0.0f;
  float s = fast_rcp_(1.0f + fast_exp_(-t));
  return 100.0f * s * (1.0f - s);
}
#endif

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

}  // namespace cnr
