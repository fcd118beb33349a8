// Ablation probe for the split-bf16 weight-gradient GEMM (256x256 tile, 16-point slabs).  Build: hipcc --offload-arch=gfx950 -O3 dw_probe.hip -o dw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int DX_JREG = 1088, DX_HALF = 4 * DX_JREG, DX_PLANE = 2 * DX_HALF, DX_OPER = 3 * DX_PLANE, DX_BUF = 2 * DX_OPER;
__device__ __forceinline__ void dx_split3(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
  h1 = (__bf16)x; float r = x - (float)h1; h2 = (__bf16)r; r = r - (float)h2; h3 = (__bf16)r;
}
// FLAGS: 1 = skip global loads after the first slab, 2 = skip MFMA, 4 = skip LDS stores (and split math), 8 = skip LDS frag reads
template <int FLAGS, int SLAB_PTS>
__global__ __launch_bounds__(512, 1) void dw_bx(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ out, long P, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const long chunk = blockIdx.x;
  const long total_slabs = (P + 15) / 16;
  const int nslab = chunk < total_slabs ? (int)((total_slabs - chunk + nchunk - 1) / nchunk) : 0;
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const bool is_y = tid >= 256;
  const int st = tid & 255;
  const int qlo = st & 1, c4 = (st >> 1) & 63, qhi = st >> 7;
  const int q = qhi * 2 + qlo;
  const float* src = (is_y ? Y : X) + c4 * 4;
  unsigned char* const sdst = smem_d + (is_y ? DX_OPER : 0) + qhi * DX_HALF + c4 * 16 + qlo * 8;
  f4 ra[4];
#define LOAD(s_) { const long pb = ((long)(s_) * nchunk + chunk) * 16 + q * 4; _Pragma("unroll") for (int i = 0; i < 4; ++i) { long pt = pb + i; if (pt >= P) pt = P - 1; ra[i] = *reinterpret_cast<const f4*>(src + pt * 256); } }
#define STORE(buf_) { unsigned char* d_ = sdst + (buf_) * DX_BUF; _Pragma("unroll") for (int j = 0; j < 4; ++j) { bf16x4 h1, h2, h3; \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { __bf16 a_, b_, c_; dx_split3(ra[i][j], a_, b_, c_); h1[i] = a_; h2[i] = b_; h3[i] = c_; } \
      *reinterpret_cast<bf16x4*>(d_ + j * DX_JREG) = h1; *reinterpret_cast<bf16x4*>(d_ + DX_PLANE + j * DX_JREG) = h2; *reinterpret_cast<bf16x4*>(d_ + 2 * DX_PLANE + j * DX_JREG) = h3; } }
  if (nslab > 0) { LOAD(0) STORE(0) }
  __syncthreads();
  const int ln = lane & 31, lh = lane >> 5;
  const int xoff = lh * DX_HALF + (ln & 3) * DX_JREG + (wr * 16 + (ln >> 2)) * 16;
  const int yoff = DX_OPER + lh * DX_HALF + (ln & 3) * DX_JREG + (wc * 32 + (ln >> 2)) * 16;
  bf16x8 a[2][3], b[3];
  for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(smem_d + xoff + p * DX_PLANE + i * 128);
  for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(smem_d + yoff + p * DX_PLANE);
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (!(FLAGS & 1) && s + 1 < nslab) LOAD(s + 1)
    const unsigned char* B_ = smem_d + buf * DX_BUF;
    if (!(FLAGS & 8)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(B_ + xoff + p * DX_PLANE + i * 128);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!(FLAGS & 8)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(B_ + yoff + p * DX_PLANE + j * 128);
      }
      if (!(FLAGS & 2)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], c, 0, 0, 0);
          acc[i][j] = c;
        }
      } else {
        for (int i = 0; i < 2; ++i) acc[i][j][0] += (float)a[i][0][0] * (float)b[0][0] + (float)a[i][1][1] * (float)b[1][1] + (float)a[i][2][2] * (float)b[2][2];
      }
    }
    if (!(FLAGS & 4) && s + 1 < nslab) STORE(buf ^ 1)
    __syncthreads();
  }
  float* o = out + chunk * 65536L;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) {
    const int kk = wc * 128 + j * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) { const int n = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); o[(long)n * 256 + kk] = acc[i][j][r] + (FLAGS ? ra[0].x * 1e-30f : 0.f); }
  }
}

template <int FLAGS>
static void run(const char* what, const float* X, const float* Y, float* out, long P, int nchunk) {
  const size_t lds = 2 * DX_BUF;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bx<FLAGS, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((dw_bx<FLAGS, 16>), dim3(nchunk), dim3(512), lds, 0, X, Y, out, P, nchunk);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((dw_bx<FLAGS, 16>), dim3(nchunk), dim3(512), lds, 0, X, Y, out, P, nchunk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-44s nchunk %4d: %.3f ms  %.1f TF/s-equiv  %.0f GB/s\n", what, nchunk, ms, 2.0 * P * 65536 / (ms * 1e-3) / 1e12, 2.0 * P * 1024 / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const long P = argc > 1 ? atol(argv[1]) : 524288;
  std::vector<float> h((size_t)P * 256);
  srand(1);
  for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
  float *X, *Y, *out;
  CK(hipMalloc(&X, h.size() * 4)); CK(hipMalloc(&Y, h.size() * 4)); CK(hipMalloc(&out, 1024 * 65536L * 4));
  CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
  CK(hipMemcpy(Y, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int nchunk : {256, 512}) {
    run<0>("full", X, Y, out, P, nchunk);
    run<1>("no global loads", X, Y, out, P, nchunk);
    run<2>("no MFMA", X, Y, out, P, nchunk);
    run<4>("no split + LDS stores", X, Y, out, P, nchunk);
    run<8>("no LDS fragment reads", X, Y, out, P, nchunk);
    run<1 | 4>("no global loads, no LDS stores", X, Y, out, P, nchunk);
    run<1 | 4 | 8>("MFMA only", X, Y, out, P, nchunk);
    run<2 | 4 | 8>("global loads only", X, Y, out, P, nchunk);
  }
  // correctness of the full variant vs float64 on a few entries
  {
    const int nchunk = 256; const size_t lds = 2 * DX_BUF;
    hipLaunchKernelGGL((dw_bx<0, 16>), dim3(nchunk), dim3(512), lds, 0, X, Y, out, P, nchunk);
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)nchunk * 65536); CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> hx((size_t)P * 256); CK(hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxv = 0;
    for (int n = 0; n < 256; n += 37) for (int k = 0; k < 256; k += 41) {
      double ref = 0; for (long p = 0; p < P; ++p) ref += (double)hx[p * 256 + n] * h[p * 256 + k];
      double got = 0; for (int c = 0; c < nchunk; ++c) got += ho[(size_t)c * 65536 + n * 256 + k];
      maxerr = fmax(maxerr, fabs(got - ref)); maxv = fmax(maxv, fabs(ref));
    }
    printf("full variant vs float64: max abs err %.3e (max |ref| %.3e, sqrt(P) = %.0f)\n", maxerr, maxv, sqrt((double)P));
  }
  return 0;
}
