// Sustained rate of v_mfma_f32_32x32x16_f16 on the whole chip: W waves per SIMD, each with A independent accumulator chains.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak_probe.hip -o mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int A>
__global__ void mfma_loop(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f32x16 acc[A];
  for (int k = 0; k < A; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int k = 0; k < A; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < A; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 12345.678f) out[0] = s;
}
template <int A>
static void run(int waves_per_simd, float* out) {
  const int iters = 2000, threads = 64 * 4 * waves_per_simd;   // one workgroup per CU
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop<A>, dim3(256), dim3(threads), 0, 0, out, iters);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mfma_loop<A>, dim3(256), dim3(threads), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double n = (double)iters * 16 * A * waves_per_simd;          // MFMAs per SIMD
  const double flop = n * 32768.0 * 1024;                            // 1024 SIMDs
  printf("%d wave(s) per SIMD, %d chain(s) per wave: %.1f ns per MFMA and SIMD, %.0f TFLOP/s (%.2f of 2500)\n", waves_per_simd, A, ms * 1e6 / n, flop / (ms * 1e-3) / 1e12,
         flop / (ms * 1e-3) / 1e12 / 2500.0);
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  run<1>(1, out); run<2>(1, out); run<4>(1, out); run<1>(2, out); run<2>(2, out); run<1>(4, out);
  return 0;
}
