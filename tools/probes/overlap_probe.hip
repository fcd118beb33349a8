// Does VALU work of one wave overlap MFMA work of the other wave on the same SIMD (gfx950)?
// MI355X: mode 1 223 ns, mode 2 334 ns, mode 3 386 ns (mostly overlapped), mode 4 422 ns per iteration.
// 512-thread workgroups, one per CU: waves 0..3 = one per SIMD ("M"), waves 4..7 = their SIMD partners ("V").
//   mode 1: M waves issue MFMAs, V waves idle      mode 2: M idle, V waves issue VALU      mode 3: both
//   mode 4: every wave interleaves MFMA and VALU in its own stream (half the MFMAs / VALU each)
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, int mode, int transc) {
  const int wave = threadIdx.x >> 6;
  const bool mwave = wave < 4;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const bool do_m = (mode == 1 && mwave) || (mode == 3 && mwave) || mode == 4;
  const bool do_v = (mode == 2 && !mwave) || (mode == 3 && !mwave) || mode == 4;
  const int n = mode == 4 ? iters / 2 : iters;
  for (int it = 0; it < n; ++it) {
    if (do_m) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    if (do_v) {
      // 96 VALU instructions per iteration: 96 x 4 cycles = the 12 MFMAs x 32 cycles of the M waves
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (transc && q == 0) v[i] = __builtin_amdgcn_exp2f(v[i]) ;
          else v[i] = v[i] * 1.0001f;
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int transc = 0; transc < 2; ++transc)
    for (int mode = 1; mode <= 4; ++mode) {
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1000, mode, transc);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, iters, mode, transc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("transc %d mode %d: %.3f ms  (%.1f ns per iteration)\n", transc, mode, ms, ms * 1e6 / iters);
    }
  return 0;
}
