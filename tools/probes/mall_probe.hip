// Does data just written by one kernel come back from the Infinity Cache (MALL) when the next kernel reads it?
// Build: hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void writer(f4* p, long n) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { f4 v = {1.f, 2.f, 3.f, (float)i}; p[i] = v; } }
__global__ void reader(const f4* p, long n, float* out) {
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { f4 v = p[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (s.x + s.y + s.z + s.w == -1.f) out[0] = s.x;
}
int main() {
  const long maxb = 2048L << 20;
  f4 *buf, *junk; float* out;
  CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&junk, maxb)); CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long mb : {16L, 32L, 64L, 128L, 192L, 256L, 384L, 512L, 1024L}) {
    const long n = (mb << 20) / 16;
    float t_hot = 0, t_cold = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n);
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, 0, buf, n, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t_hot += ms;
      hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n);
      hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, junk, maxb / 16);   // evict
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, 0, buf, n, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); t_cold += ms;
    }
    printf("%5ld MB: read right after write %.1f GB/s, read after 2 GB of other writes %.1f GB/s\n", mb, (mb << 20) / (t_hot / 5 * 1e-3) / 1e9, (mb << 20) / (t_cold / 5 * 1e-3) / 1e9);
  }
  return 0;
}
