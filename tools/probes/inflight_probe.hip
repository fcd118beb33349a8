// How much of the 5.2 TB/s memory-side line of the layer kernel is a matter of bytes in flight?  One workgroup of 512 threads per CU (120 KB of
// LDS keep a second one away), 32 KB tiles (32 rows x 1 KB) read with D tiles in flight per workgroup and written back, one barrier per tile
// like the layer kernel.  Build: hipcc --offload-arch=gfx950 -O3 inflight_probe.hip -o inflight_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(512, 1) void stream(const f4* __restrict__ in, f4* __restrict__ out, long ntiles, int tpw) {
  extern __shared__ float lds[];
  const long t0 = (long)blockIdx.x * tpw, tl = t0 + tpw - 1;
  const int srow = threadIdx.x >> 4, sc = threadIdx.x & 15;       // 16 threads per row, 4 x 16 B each (the layer kernel's staging map)
  f4 r[D][4];
#define LD(S_, t_) { const long tq = (t_) < tl ? (t_) : tl; const f4* p = in + (tq * 32 + srow) * 64 + sc; \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) r[S_][i] = p[i * 16]; }
#define ST(S_, t_) { f4* p = out + ((t_) * 32 + (threadIdx.x & 63) / 8 + 8 * (threadIdx.x >> 7)) * 64 + (threadIdx.x >> 6 & 1) * 32 + (threadIdx.x & 7); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { f4 v = r[S_][i]; v.x += 1.0f; p[i * 8] = v; } }
#pragma unroll
  for (int d = 0; d < D; ++d) LD(d, t0 + d)
  for (long t = t0; t <= tl; t += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (t + d <= tl) ST(d, t + d)        // (consumes set d: the wait leaves the younger sets in flight)
      LD(d, t + d + D)
      asm volatile("s_barrier" ::: "memory");
    }
  }
}
template <int D>
static void run(const f4* in, f4* out, long ntiles) {
  const int tpw = (int)(ntiles / 256);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream<D>, dim3(256), dim3(512), 120 * 1024, 0, in, out, ntiles, tpw);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream<D>, dim3(256), dim3(512), 120 * 1024, 0, in, out, ntiles, tpw);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; CK(hipGetLastError());
  printf("%d tile(s) of 32 KB in flight per CU: %.3f ms, %.0f GB/s (read + write)\n", D, ms, 2.0 * ntiles * 32768 / (ms * 1e-3) / 1e9);
}
int main() {
  const long ntiles = 16384;
  f4 *in, *out; CK(hipMalloc(&in, ntiles * 32768)); CK(hipMalloc(&out, ntiles * 32768)); CK(hipMemset(in, 0, ntiles * 32768));
  run<1>(in, out, ntiles); run<2>(in, out, ntiles); run<3>(in, out, ntiles); run<4>(in, out, ntiles); run<6>(in, out, ntiles); run<8>(in, out, ntiles);
  return 0;
}
