// What does the matrix pipe itself cost in watts?  v_mfma_f32_32x32x16_f16 on the whole chip for several seconds (so that rocm-smi can sample the board),
// two waves per SIMD, two accumulator chains per wave, operands either CONSTANT (the registers never change: the data paths barely toggle) or RANDOM
// (every MFMA takes another pair of 8 random fragments held in registers).  Prints the sustained rate per phase; the caller samples power / clock
// (tools/mfma_power.py).  Build: hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int RANDOM>
__global__ __launch_bounds__(512, 1) void mfma_loop(const _Float16* src, float* out, int iters) {
  f16x8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      a[q][i] = RANDOM ? src[((q * 8 + i) * 512 + threadIdx.x) & 32767] : (_Float16)0.5f;
      b[q][i] = RANDOM ? src[((q * 8 + i + 32) * 512 + threadIdx.x) & 32767] : (_Float16)0.25f;
    }
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + k) & 3], b[(u >> 2) & 3], acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 12345.678f) out[0] = s;
}

template <int RANDOM>
static void run(const _Float16* src, float* out, double seconds, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop<RANDOM>, dim3(wgs), dim3(512), 0, 0, src, out, 100);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  double ms_tot = 0; long launches = 0;
  printf("phase %s start\n", RANDOM ? "random" : "constant"); fflush(stdout);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_loop<RANDOM>, dim3(wgs), dim3(512), 0, 0, src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_tot += ms; launches += 4;
  }
  const double n = (double)launches * iters * 16 * 2 * 2;           // MFMAs per SIMD (2 waves x 2 chains)
  const double flop = n * 32768.0 * 4 * wgs;
  printf("phase %s end: %.1f ns per MFMA and SIMD, %.0f TFLOP/s of f16 MFMA (%.2f of 2500) over %.1f s\n", RANDOM ? "random" : "constant", ms_tot * 1e6 / n,
         flop / (ms_tot * 1e-3) / 1e12, flop / (ms_tot * 1e-3) / 1e12 / 2500.0, ms_tot * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  const int wgs = argc > 2 ? atoi(argv[2]) : 256;   // 256 = one workgroup per CU (the whole chip); 1 = a single CU (no power effect: the issue rate alone)
  std::vector<_Float16> h(32768);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.0f - 1.0f);
  _Float16* src; float* out;
  CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&out, 4));
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  run<0>(src, out, seconds, wgs);
  run<1>(src, out, seconds, wgs);
  return 0;
}
