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

struct alignas(16) f4 {
  float x, y, z, w;
};

CNR_HD float& f4_at(f4& v, int i) { return (&v.x)[i]; }

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr int kMaxLayers = 12;   // per MLP
constexpr int kEmb = 40;         // padded width of the SDF positional-encoding buffer (39 -> 40)
constexpr int kAux = 40;         // padded width of the auxiliary input buffer [p(3) g(3) PE4(dir)(27) pad]

CNR_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// nn.Softplus(beta=100, threshold=20)                                      (reference fields.py:77)
CNR_HD float softplus100(float z) {
  float t = 100.0f * z;
  return t > 20.0f ? z : log1pf(expf(t)) * 0.01f;
}
// d softplus / dz = sigmoid(100 z) (1 above the threshold)
CNR_HD float softplus100_d1(float z) {
  float t = 100.0f * z;
  return t > 20.0f ? 1.0f : sigmoidf_(t);
}
// d2 softplus / dz2 = 100 s (1-s) (0 above the threshold)
CNR_HD float softplus100_d2(float z) {
  float t = 100.0f * z;
  if (t > 20.0f) return 0.0f;
  float s = sigmoidf_(t);
  return 100.0f * s * (1.0f - s);
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

}  // namespace cnr
