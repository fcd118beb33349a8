// Can consecutive layers hand their tiles over through the L2 of one XCD instead of through the fabric?
// 256 workgroups (one per CU: 120 KB of LDS each), workgroup b -> XCD b % 8 (checked with XCC_ID), stage (b / 8) % D of pipeline
// (b % 8) + 8 * (b / (8 * D)).  Stage s waits for the flag of stage s - 1, reads the 32 KB tile its producer wrote, spends `busy`
// MFMA-ish cycles, writes its own tile and raises its flag.  Compared with D separate launches that each read and write HBM.
// Build: hipcc --offload-arch=gfx950 -O3 pipe_probe.hip -o pipe_probe       Run: ./pipe_probe [D] [busy_iters] [fence_mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TILE_F4 = 2048;   // 32 KB per tile = 512 threads x 4 x 16 B
constexpr int MAXD = 8;

struct Args {
  f4* buf[MAXD + 1];   // buf[0] = source, buf[s + 1] = output of stage s
  int* flags;          // [npipes][MAXD] tiles finished
  int* info;           // [grid][2] xcc id, hw id
  int* err;
  long ntiles;         // tiles per pipeline (pipelined) / total tiles (flat)
  int D, busy, fence_mode;
};

__device__ __forceinline__ float burn(float x, int iters) {
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
  return x;
}

struct Tile4 { f4 v0, v1, v2, v3; };
__device__ __forceinline__ Tile4 tile_load(const f4* __restrict__ in, long tile) {
  const f4* src = in + tile * TILE_F4 + threadIdx.x;
  Tile4 t; t.v0 = src[0]; t.v1 = src[512]; t.v2 = src[1024]; t.v3 = src[1536];
  return t;
}
__device__ __forceinline__ void tile_finish(Tile4 t, f4* __restrict__ out, long tile, int busy) {
  const float k = burn(1.0f + t.v0.y * 1e-30f, busy);
  t.v0.x += k; t.v1.x += k; t.v2.x += k; t.v3.x += k;
  f4* dst = out + tile * TILE_F4 + threadIdx.x;
  dst[0] = t.v0; dst[512] = t.v1; dst[1024] = t.v2; dst[1536] = t.v3;
}

// Software-pipelined stage.  Iteration j: (thread 0) the producer's flag covers tile j + 1 -> s_waitcnt vmcnt(4) [loads(j) and
// stores(j - 2) have completed, stores(j - 1) may still be in flight] -> barrier -> raise the own flag to j - 1 -> issue loads(j + 1)
// -> compute + store tile j -> refresh the cached producer flag with a scalar load (lgkmcnt: does not disturb the vmcnt order).
// PIPE = false: the same loop without flags (one launch per layer).
__device__ __forceinline__ void flag_request(const int* p, int& v) { asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(v) : "s"(p) : "memory"); }
__device__ __forceinline__ void flag_arrive(int& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) :: "memory"); }

template <bool PIPE>
__device__ __forceinline__ void stage_loop(const f4* in, f4* out, long tile0, long n, int busy, int* myflag, const int* upflag, int* err) {
  const bool wait_up = PIPE && upflag != nullptr;
  const bool t0 = __builtin_amdgcn_readfirstlane(threadIdx.x) == 0;   // wave 0 (uniform)
  int seen = 0;
  bool pending = false;
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  auto need = [&](long tiles) {   // blocks wave 0 until the producer has completed `tiles` tiles
    if (!wait_up || !t0) return;
    if (pending) { flag_arrive(seen); pending = false; }
    int spins = 0;
    while (seen < (int)tiles) {
      flag_request(upflag, seen); flag_arrive(seen);
      if (seen >= (int)tiles) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 2000000) { if (threadIdx.x == 0) { bad = 1; atomicAdd(err, 1); } break; }
    }
  };
  need(1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (bad) return;
  Tile4 cur = tile_load(in, tile0), nxt = cur;
  for (long j = 0; j < n; ++j) {
    const bool more = j + 1 < n;
    if (more) need(j + 2);
    if (more && j > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (bad) return;
    if (PIPE && threadIdx.x == 0 && j >= 2) __hip_atomic_store(myflag, (int)(j - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (more) nxt = tile_load(in, tile0 + j + 1);
    tile_finish(cur, out, tile0 + j, busy);
    if (wait_up && t0 && !pending && seen < (int)n) { flag_request(upflag, seen); pending = true; }
    cur = nxt;
  }
  if (PIPE) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store(myflag, (int)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(512, 1) void pipe_kernel(const Args a) {
  extern __shared__ float lds[];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    a.info[2 * b] = (int)xcc; a.info[2 * b + 1] = (int)hw;
  }
  const int xcd = b & 7, idx = b >> 3;
  const int stage = idx % a.D, pipe = xcd + 8 * (idx / a.D);
  stage_loop<true>(a.buf[stage], a.buf[stage + 1], (long)pipe * a.ntiles, a.ntiles, a.busy, a.flags + pipe * MAXD + stage,
                   stage > 0 ? a.flags + pipe * MAXD + stage - 1 : nullptr, a.err);
}

__global__ __launch_bounds__(512, 1) void flat_kernel(const f4* in, f4* out, long ntiles, int busy) {
  extern __shared__ float lds[];
  const long per = (ntiles + gridDim.x - 1) / gridDim.x;
  long t0 = (long)blockIdx.x * per, t1 = t0 + per;
  if (t1 > ntiles) t1 = ntiles;
  if (t0 < t1) stage_loop<false>(in, out, t0, t1 - t0, busy, nullptr, nullptr, nullptr);
}

int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 8;
  const int busy = argc > 2 ? atoi(argv[2]) : 2000;
  const int fence_mode = argc > 3 ? atoi(argv[3]) : 0;
  const long total_tiles = 16384;   // 512 MB per array
  const int npipes = 256 / D;
  Args a{};
  a.D = D; a.busy = busy; a.fence_mode = fence_mode; a.ntiles = total_tiles / npipes;
  for (int s = 0; s <= D; ++s) CK(hipMalloc(&a.buf[s], total_tiles * TILE_F4 * sizeof(f4)));
  CK(hipMalloc(&a.flags, 256 * MAXD * sizeof(int)));
  CK(hipMalloc(&a.info, 256 * 2 * sizeof(int)));
  CK(hipMalloc(&a.err, sizeof(int)));
  CK(hipMemset(a.buf[0], 0, total_tiles * TILE_F4 * sizeof(f4)));
  CK(hipMemset(a.err, 0, sizeof(int)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&flat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_flat = 0, ms_pipe = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    for (int s = 0; s < D; ++s) hipLaunchKernelGGL(flat_kernel, dim3(256), dim3(512), 120 * 1024, 0, a.buf[s], a.buf[s + 1], total_tiles, busy);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ms_flat += ms;
  }
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(a.flags, 0, 256 * MAXD * sizeof(int), 0));
    for (int s = 1; s <= D; ++s) CK(hipMemsetAsync(a.buf[s], 0xff, total_tiles * TILE_F4 * sizeof(f4), 0));   // stale data = NaN
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(pipe_kernel, dim3(256), dim3(512), 120 * 1024, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ms_pipe += ms;
  }
  CK(hipDeviceSynchronize());
  int err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
  std::vector<int> info(512); CK(hipMemcpy(info.data(), a.info, 512 * 4, hipMemcpyDeviceToHost));
  int rr_ok = 1; for (int b = 0; b < 256; ++b) if ((info[2 * b] & 0xf) != (b & 7)) rr_ok = 0;
  printf("xcc of workgroups 0..15:"); for (int b = 0; b < 16; ++b) printf(" %d", info[2 * b] & 0xf); printf("   round-robin mapping holds: %s\n", rr_ok ? "yes" : "NO");
  // verify: every first float of every f4 lane-0 element should be D * k
  std::vector<f4> chk(TILE_F4); long bad = 0;
  for (long t : {0L, 1L, total_tiles / 2, total_tiles - 1}) {
    CK(hipMemcpy(chk.data(), a.buf[D] + t * TILE_F4, TILE_F4 * sizeof(f4), hipMemcpyDeviceToHost));
    for (int i = 0; i < TILE_F4; ++i) if (!(chk[i].x > 0.5f * D && chk[i].x < 2.0f * D)) ++bad;
  }
  const double gb = (double)total_tiles * TILE_F4 * 16 / 1e9;
  printf("D=%d busy=%d fence=%d: %d flat launches %.3f ms (%.0f GB/s of read+write), pipelined %.3f ms (fabric bytes if L2 hand-over works: %.0f GB/s), spin time-outs %d, bad values %ld\n",
         D, busy, fence_mode, D, ms_flat / 3, 2 * D * gb / (ms_flat / 3 * 1e-3), ms_pipe / 3, (D + 1) * gb / (ms_pipe / 3 * 1e-3), err, bad);
  return 0;
}
