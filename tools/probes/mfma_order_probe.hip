// Does the ORDER of the three MFMAs of a split product (a1 w2, a2 w1, a1 w1) matter for the power of the matrix pipe?  In the shipped order both
// operands change between the first and the second MFMA; in the "Gray" order (a1 w2 -> a1 w1 -> a2 w1) exactly one operand changes per step.
// v_mfma_f32_32x32x16_f16 on the whole chip, two waves per SIMD, random f16 operands in 4 register sets (k16 blocks), one accumulator per wave.
//   mode 0: shipped order   mode 1: Gray order   mode 2: both operands change at every MFMA   mode 3: constant operands
// Prints the sustained rate per phase; tools/mfma_order.py samples rocm-smi.  Build: hipcc --offload-arch=gfx950 -O3 mfma_order_probe.hip -o mfma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void mfma_loop(const _Float16* src, float* out, int iters) {
  f16x8 a1[4], a2[4], w1[4], w2[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      const int o = (q * 8 + i) * 512 + threadIdx.x;
      a1[q][i] = MODE == 3 ? (_Float16)0.5f : src[o & 32767];
      a2[q][i] = MODE == 3 ? (_Float16)0.5f : (_Float16)((float)src[(o + 7919) & 32767] * 0.0004f);    // the low plane: ~2^-11 of the high one
      w1[q][i] = MODE == 3 ? (_Float16)0.25f : src[(o + 104729) & 32767];
      w2[q][i] = MODE == 3 ? (_Float16)0.25f : (_Float16)((float)src[(o + 1299709) & 32767] * 0.0004f);
    }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (MODE == 0 || MODE == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w2[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[kb], w1[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w1[kb], acc, 0, 0, 0);
      } else if (MODE == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w2[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w1[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[kb], w1[kb], acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w2[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[(kb + 1) & 3], w1[(kb + 2) & 3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[(kb + 3) & 3], w1[kb], acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.678f) out[0] = s;
}

// the same three-MFMA pattern on v_mfma_f32_16x16x32_f16 (half the FLOP per instruction, a quarter of the accumulator registers): is the other f16 shape cheaper per FLOP?
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 1) void mfma_loop_16(const _Float16* src, float* out, int iters) {
  f16x8 a1[4], a2[4], w1[4], w2[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      const int o = (q * 8 + i) * 512 + threadIdx.x;
      a1[q][i] = src[o & 32767];
      a2[q][i] = (_Float16)((float)src[(o + 7919) & 32767] * 0.0004f);
      w1[q][i] = src[(o + 104729) & 32767];
      w2[q][i] = (_Float16)((float)src[(o + 1299709) & 32767] * 0.0004f);
    }
  f32x4 acc[2];
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[kb], w2[(kb + k) & 3], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[kb], w1[(kb + k) & 3], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[kb], w1[(kb + k) & 3], acc[k], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 4; ++i) s += acc[k][i];
  if (s == 12345.678f) out[0] = s;
}
static void run16(const _Float16* src, float* out, double seconds, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop_16, dim3(wgs), dim3(512), 0, 0, src, out, 100);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  double ms_tot = 0; long launches = 0;
  printf("phase mode4 start\n"); fflush(stdout);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_loop_16, dim3(wgs), dim3(512), 0, 0, src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_tot += ms; launches += 4;
  }
  const double n = (double)launches * iters * 24 * 2;           // MFMAs per SIMD (2 waves x 24 per iteration)
  const double flop = n * 16384.0 * 4 * wgs;
  printf("phase mode4 end: %.1f ns per MFMA and SIMD, %.0f TFLOP/s of f16 MFMA over %.1f s\n", ms_tot * 1e6 / n, flop / (ms_tot * 1e-3) / 1e12, ms_tot * 1e-3);
  fflush(stdout);
}

// the shipped order on v_mfma_f32_32x32x16_f16 with TWO accumulator chains per wave (the weight-gradient waves have eight): the fair partner of mode 4
__global__ __launch_bounds__(512, 1) void mfma_loop_32x2(const _Float16* src, float* out, int iters) {
  f16x8 a1[4], a2[4], w1[4], w2[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      const int o = (q * 8 + i) * 512 + threadIdx.x;
      a1[q][i] = src[o & 32767];
      a2[q][i] = (_Float16)((float)src[(o + 7919) & 32767] * 0.0004f);
      w1[q][i] = src[(o + 104729) & 32767];
      w2[q][i] = (_Float16)((float)src[(o + 1299709) & 32767] * 0.0004f);
    }
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w2[(kb + k) & 3], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[kb], w1[(kb + k) & 3], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], w1[(kb + k) & 3], acc[k], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 12345.678f) out[0] = s;
}
static void run32x2(const _Float16* src, float* out, double seconds, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop_32x2, dim3(wgs), dim3(512), 0, 0, src, out, 100);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  double ms_tot = 0; long launches = 0;
  printf("phase mode5 start\n"); fflush(stdout);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_loop_32x2, dim3(wgs), dim3(512), 0, 0, src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_tot += ms; launches += 4;
  }
  const double n = (double)launches * iters * 24 * 2;
  const double flop = n * 32768.0 * 4 * wgs;
  printf("phase mode5 end: %.1f ns per MFMA and SIMD, %.0f TFLOP/s of f16 MFMA over %.1f s\n", ms_tot * 1e6 / n, flop / (ms_tot * 1e-3) / 1e12, ms_tot * 1e-3);
  fflush(stdout);
}

template <int MODE>
static void run(const _Float16* src, float* out, double seconds, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop<MODE>, dim3(wgs), dim3(512), 0, 0, src, out, 100);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  double ms_tot = 0; long launches = 0;
  printf("phase mode%d start\n", MODE); fflush(stdout);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(wgs), dim3(512), 0, 0, src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_tot += ms; launches += 4;
  }
  const double n = (double)launches * iters * 12 * 2;           // MFMAs per SIMD (2 waves x 12 per iteration)
  const double flop = n * 32768.0 * 4 * wgs;
  printf("phase mode%d end: %.1f ns per MFMA and SIMD, %.0f TFLOP/s of f16 MFMA over %.1f s\n", MODE, ms_tot * 1e6 / n, flop / (ms_tot * 1e-3) / 1e12, ms_tot * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  const int wgs = argc > 2 ? atoi(argv[2]) : 256;
  std::vector<_Float16> h(32768);
  srand(1);
  for (auto& v : h) v = (_Float16)(((rand() / (float)RAND_MAX) * 2.0f - 1.0f) * 16384.0f);   // the top f16 binades, like a scaled row
  _Float16* src; float* out;
  CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&out, 4));
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  if (argc > 3) {   // shape comparison only: two accumulator chains each, alternating
    for (int rep = 0; rep < 3; ++rep) { run32x2(src, out, seconds, wgs); run16(src, out, seconds, wgs); }
    return 0;
  }
  run<3>(src, out, seconds, wgs);
  run<0>(src, out, seconds, wgs);
  run<1>(src, out, seconds, wgs);
  run<2>(src, out, seconds, wgs);
  run<0>(src, out, seconds, wgs);
  run16(src, out, seconds, wgs);
  run<0>(src, out, seconds, wgs);
  run16(src, out, seconds, wgs);
  return 0;
}
