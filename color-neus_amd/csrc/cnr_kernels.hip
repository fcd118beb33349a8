// HIP kernels for gfx950 (MI355X / CDNA4).  No CUDA compatibility layer, no dual paths: wave = 64 lanes,
// FP32 MFMA (v_mfma_f32_32x32x2_f32) for every fully-connected layer, LDS-staged tiles, per-ray kernels
// with one wavefront per ray and shuffle scans.
//
// Kernels
//   layer_gemm_kernel<NT>   C[P x N] = epilogue(A[P x K] * W[N x K]^T), 128-point tile x full N per workgroup,
//                           K streamed in 16-wide slabs through double-buffered LDS, fused prologue (operand view)
//                           and fused epilogue (bias / softplus' / relu mask / second-order terms / split stores)
//   dw_gemm_kernel<...>     dW[N x K] = sum_pts X[pt][n] * Y[pt][k]  (weight gradients; reduce dimension = points)
//   upsample / merge / composite_fwd / composite_bwd: one wavefront per ray, ray state staged in LDS
//   point-wise kernels: positional encodings, PE Jacobians, weight preparation (weight-norm) and its backward
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <vector>

#include "cnr_backend.h"
#include "cnr_loss.h"
#include "cnr_bodies.h"
#include "cnr_hip_util.h"

namespace cnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

hipError_t g_first_error = hipSuccess;
const char* g_first_error_where = "";

const char* be_name() { return "hip-gfx950"; }

// ---- per-launch timing (HIP events recorded on the launch stream) ---------------------------------------------------
struct TimingRec { KernelTiming t; hipEvent_t e0, e1; };
static bool g_timing_on = false;
static std::vector<TimingRec> g_timing;
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t get_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void timing_begin(const char* name, int kind, int nt, long P, int N, int K, int pairs, hipStream_t s, double bytes) {
  if (!g_timing_on) return;
  TimingRec r;
  snprintf(r.t.name, sizeof r.t.name, "%s", name);
  r.t.kind = kind; r.t.nt = nt; r.t.P = P; r.t.N = N; r.t.K = K; r.t.pairs = pairs; r.t.ms = 0.f; r.t.bytes = bytes;
  r.e0 = get_event(); r.e1 = get_event();
  (void)hipEventRecord(r.e0, s);
  g_timing.push_back(r);
}
void timing_end(hipStream_t s) {
  if (!g_timing_on || g_timing.empty()) return;
  (void)hipEventRecord(g_timing.back().e1, s);
}
void be_timing_enable(int on) { g_timing_on = on != 0; }
int be_timing_collect(KernelTiming* out, int max_records) {
  int n = 0;
  for (auto& r : g_timing) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    r.t.ms = ms;
    if (out && n < max_records) out[n] = r.t;
    ++n;
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_timing.clear();
  return n;
}

namespace {
struct Roctx {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
  Roctx() {
    if (!debug_flags().roctx) return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void be_range_push(const char* name) { if (roctx().push) (void)roctx().push(name); }
void be_range_pop() { if (roctx().pop) (void)roctx().pop(); }

int be_check_last_error(char* msg, size_t n) {
  if (g_first_error == hipSuccess) return 0;
  snprintf(msg, n, "HIP error in %s: %s", g_first_error_where, hipGetErrorString(g_first_error));
  g_first_error = hipSuccess;
  return -1;
}

void be_memset_zero(void* p, size_t bytes, cnr_stream s) {
  hipError_t e = hipMemsetAsync(p, 0, bytes, s);
  if (e != hipSuccess && g_first_error == hipSuccess) { g_first_error = e; g_first_error_where = "memset"; }
}

__global__ void zero_cols_kernel(float* p, int ld, int c0, int c1, long rows) {
  const int w = c1 - c0;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * w) p[(i / w) * ld + c0 + (int)(i % w)] = 0.0f;
}
void be_zero_cols(float* p, int ld, int c0, int c1, long rows, cnr_stream s) {
  if (c1 <= c0 || rows <= 0) return;
  const long n = rows * (c1 - c0);
  hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, ld, c0, c1, rows);
  CNR_LAUNCH_CHECK("zero_cols");
}

__global__ __launch_bounds__(256) void copy_cols_kernel(float* dst, int ld_dst, const float* src, int ld_src, int ncols, long rows) {
  const long n = rows * ncols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / ncols;
    const int c = (int)(i - r * ncols);
    dst[r * ld_dst + c] = src[r * ld_src + c];
  }
}
void be_copy_cols(float* dst, int ld_dst, const float* src, int ld_src, int ncols, long rows, cnr_stream s) {
  if (ncols <= 0 || rows <= 0) return;
  long blocks = (rows * ncols + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, ld_dst, src, ld_src, ncols, rows);
  CNR_LAUNCH_CHECK("copy_cols");
}

// ------------------------------------------------------------------------------------------------
// backward of a 3-wide head: one pass over the 1 KB/point input of the head (see HeadBwd in cnr_backend.h).  One thread per column,
// points of a slot in order (fixed summation order), 8 points in flight.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void head_bwd_kernel(const HeadBwd p, long pts_per_slot) {
  // 64 column groups (4 columns, 16-byte accesses) x 8 row lanes; a row lane walks the points p0 + rl, p0 + rl + 8, ... of the slot, eight of
  // them in flight; the row lanes are folded in a fixed order through LDS at the end
  __shared__ float red[8][4][256];
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6, k = cg * 4;
  const long p0 = (long)blockIdx.x * pts_per_slot;
  long p1 = p0 + pts_per_slot;
  if (p1 > p.P) p1 = p.P;
  const bool live = k < p.K;                      // (K is a multiple of 4 here: 256)
  f4 w[4], acc[4];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) { acc[j] = z4; w[j] = (live && j < p.n) ? *reinterpret_cast<const f4*>(p.W + (long)j * p.ldw + k) : z4; }
  // the rows of the next trip are requested before this trip's arithmetic and STORES: a load issued behind the stores would be waited for
  // together with the wave's whole store queue (one in-order counter), and one workgroup per CU has nothing else to hide the latency with
  f4 an[8], dn[8];
  const int kl = live ? k : 0;
  auto fetch = [&](long pt) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      long q = pt + 8 * u < p1 ? pt + 8 * u : p1 - 1;
      if (q < 0) q = 0;
      an[u] = *reinterpret_cast<const f4*>(p.aux + q * p.ldaux + kl);   // (every lane loads from a clamped address; dead column groups are masked below)
      dn[u] = *reinterpret_cast<const f4*>(p.dtop + q * p.ldt);     // (ldt is a multiple of 4; columns >= n are zero padding)
    }
  };
  fetch(p0 + rl);
  for (long pt = p0 + rl; pt < p1; pt += 64) {
    f4 a[8], d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = live ? an[u] : z4; d[u] = dn[u]; }
    fetch(pt + 64);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (pt + 8 * u < p1) {
        const float dj[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
        f4 v = z4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v.x = fmaf(w[j].x, dj[j], v.x); v.y = fmaf(w[j].y, dj[j], v.y); v.z = fmaf(w[j].z, dj[j], v.z); v.w = fmaf(w[j].w, dj[j], v.w);
          acc[j].x = fmaf(dj[j], a[u].x, acc[j].x); acc[j].y = fmaf(dj[j], a[u].y, acc[j].y);
          acc[j].z = fmaf(dj[j], a[u].z, acc[j].z); acc[j].w = fmaf(dj[j], a[u].w, acc[j].w);
          cs[j] += dj[j];
        }
        f4 o;
        o.x = a[u].x > 0.0f ? v.x : 0.0f; o.y = a[u].y > 0.0f ? v.y : 0.0f; o.z = a[u].z > 0.0f ? v.z : 0.0f; o.w = a[u].w > 0.0f ? v.w : 0.0f;
        if (live) *reinterpret_cast<f4*>(p.dout + (pt + 8 * u) * p.ldo + k) = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f4*>(&red[rl][j][k]) = acc[j];
  __shared__ float redc[8][4];
  if (cg == 0) { for (int j = 0; j < 4; ++j) redc[rl][j] = cs[j]; }
  __syncthreads();
  float* out = p.partial + (long)blockIdx.x * p.npad * p.ldk;
  for (int e = threadIdx.x; e < p.n * 256; e += 512) {
    const int j = e >> 8, c = e & 255;
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += red[r][j][c];
    if (c < p.ldk) out[(long)j * p.ldk + c] = sum;
  }
  if (p.colsum && threadIdx.x < p.n) {
    float sum = 0.0f;
    for (int r = 0; r < 8; ++r) sum += redc[r][threadIdx.x];
    p.colsum[(long)blockIdx.x * p.npad + threadIdx.x] = sum;
  }
}
void be_head_bwd(const HeadBwd& p, cnr_stream s) {
  const long per = round_up((int)((p.P + p.nslots - 1) / p.nslots), 64);
  TimingScope ts_("head_bwd", 2, p.n, p.P, p.K, p.n, 1, s, (double)p.P * (8.0 * p.K + 16.0));
  hipLaunchKernelGGL(head_bwd_kernel, dim3(p.nslots), dim3(512), 0, s, p, per);
  CNR_LAUNCH_CHECK("head_bwd");
}

// ------------------------------------------------------------------------------------------------
// strips of a (256 + nt)-input layer in the backward pass (see StripBwd in cnr_backend.h): one pass over the 1 KB/point cotangent.
// A wave owns one point at a time (lane = 4-column group), PTS points in flight.  The PTS x NT partial dot products of a lane are summed
// over the 64 lanes by a halving exchange (each step a lane gives away the half of its values that belongs to the other side of the
// mask): 63 shuffles for 64 sums, lane L ends with the sum of index L = point (L / NT), column (L % NT).  Fixed order: deterministic.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512) void strip_bwd_kernel(const StripBwd p, long pts_per_slot) {
  constexpr int PTS = 64 / NT;                    // points in flight per wave
  extern __shared__ __attribute__((aligned(16))) float strip_red[];   // [8 waves][NT][256]
  const int lane = threadIdx.x & 63, k = lane * 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long p0 = (long)blockIdx.x * pts_per_slot;
  long p1 = p0 + pts_per_slot;
  if (p1 > p.P) p1 = p.P;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  f4 w[NT], acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { acc[j] = z4; w[j] = j < p.nt ? *reinterpret_cast<const f4*>(p.Wt + (long)(256 + j) * p.ldwt + k) : z4; }
  const int yu = lane / NT, yj = lane % NT;       // the (point, column) whose y value this lane fetches and whose dot product it ends up with
  // the rows of the next trip are requested before this trip's arithmetic (one workgroup of 8 waves per CU: nothing else hides the latency)
  f4 dn[PTS];
  float yn = 0.0f;
  auto fetch = [&](long pt) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
      long q = pt + u < p1 ? pt + u : p1 - 1;
      if (q < p0) q = p0 < p.P ? p0 : p.P - 1;          // (an empty slot: never used)
      dn[u] = *reinterpret_cast<const f4*>(p.dout + q * p.ldo + k);
    }
    long qy = pt + yu < p1 ? pt + yu : p1 - 1;
    if (qy < 0) qy = 0;
    yn = p.y[qy * p.ldy + (yj < p.nt ? yj : 0)];        // (every lane loads from a clamped address; masked below)
  };
  fetch(p0 + (long)wv * PTS);
  for (long pt = p0 + (long)wv * PTS; pt < p1; pt += 8 * PTS) {
    f4 d[PTS];
#pragma unroll
    for (int u = 0; u < PTS; ++u) d[u] = dn[u];
    const bool mine = pt + yu < p1 && yj < p.nt;
    const float yl = mine ? yn : 0.0f;
    fetch(pt + 8 * PTS);
    float v[64];
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float yv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yl), u * NT + j));   // 0 past the end / j >= nt
        acc[j].x = fmaf(yv, d[u].x, acc[j].x); acc[j].y = fmaf(yv, d[u].y, acc[j].y);
        acc[j].z = fmaf(yv, d[u].z, acc[j].z); acc[j].w = fmaf(yv, d[u].w, acc[j].w);
        v[u * NT + j] = fmaf(d[u].w, w[j].w, fmaf(d[u].z, w[j].z, fmaf(d[u].y, w[j].y, d[u].x * w[j].x)));
      }
    }
    // halving exchange.  Steps 32 and 16 by VALU lane swaps: v_permlane32_swap(v[i], v[i + mk]) leaves {(lo of v[i], lo of v[i + mk]), (hi of v[i], hi of v[i + mk])},
    // i.e. in every lane the value it keeps and the one its partner sends -- their sum is keep + recv of the shuffle form (a + b commutes: same bits)
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 32]), false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int mk = 8; mk >= 1; mk >>= 1) {
      const bool up = (lane & mk) != 0;
#pragma unroll
      for (int i = 0; i < mk; ++i) {
        const float keep = up ? v[i + mk] : v[i];
        const float send = up ? v[i] : v[i + mk];
        v[i] = keep + __shfl_xor(send, mk);
      }
    }
    if (p.tail && mine) p.tail[(pt + yu) * p.ldt + yj] = v[0] * p.tail_scale;
  }
  // fold the 8 waves in a fixed order; a slot row is [ .. 256 main columns (another launch) .. | nt strip columns | zero pad up to ldk ]
#pragma unroll
  for (int j = 0; j < NT; ++j) *reinterpret_cast<f4*>(strip_red + ((long)wv * NT + j) * 256 + k) = acc[j];
  __syncthreads();
  float* out = p.partial + (long)blockIdx.x * p.npad * p.ldk;
  const int wcols = p.ldk - 256;
  for (int e = threadIdx.x; e < 256 * wcols; e += 512) {
    const int n = e / wcols, j = e - n * wcols;
    float sum = 0.0f;
    if (j < p.nt) {
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += strip_red[((long)r * NT + j) * 256 + n];
    }
    if (n < p.npad) out[(long)n * p.ldk + 256 + j] = sum;
  }
}
// forward of a narrow head (see HeadFwd): a wave owns 16 points at a time (lane = 4-column group), the 16 x 4 partial dot products of a lane
// are summed over the 64 lanes by the halving exchange of strip_bwd_kernel; lane L then applies the head's epilogue to point L / 4, column L % 4
// (and to the pad columns 4.., which epi_apply zero-fills where the epilogue kind asks for it).
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadFwd p) {
  const int lane = threadIdx.x & 63, k = lane * 4;
  const long P = p.P_dev ? (long)*p.P_dev : p.P;
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  f4 w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = j < p.n ? *reinterpret_cast<const f4*>(p.W + (long)j * p.ldw + k) : z4;
  const long wave_id = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (long)gridDim.x * 4;
  for (long pt = wave_id * 16; pt < P; pt += nwaves * 16) {
    f4 d[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const long q = pt + u < P ? pt + u : P - 1;
      d[u] = *reinterpret_cast<const f4*>(p.h + q * p.ldh + k);
    }
    float v[64];
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[u * 4 + j] = fmaf(d[u].w, w[j].w, fmaf(d[u].z, w[j].z, fmaf(d[u].y, w[j].y, d[u].x * w[j].x)));
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) {
      const bool up = (lane & mk) != 0;
#pragma unroll
      for (int i = 0; i < mk; ++i) {
        const float keep = up ? v[i + mk] : v[i];
        const float send = up ? v[i] : v[i + mk];
        v[i] = keep + __shfl_xor(send, mk);
      }
    }
    const long row = pt + (lane >> 2);
    if (row < P) {
      const int j = lane & 3;
      epi_apply(p.E, row, j, v[0]);
      epi_apply(p.E, row, j + 4, 0.0f);
      epi_apply(p.E, row, j + 8, 0.0f);
      epi_apply(p.E, row, j + 12, 0.0f);
    }
  }
}
void be_head_fwd(const HeadFwd& p, cnr_stream s) {
  if (p.P <= 0) return;
  long blocks = (p.P + 63) / 64;          // 4 waves x 16 points per block and trip
  if (blocks > 2048) blocks = 2048;
  TimingScope ts_("head_fwd", 2, p.n, p.P, p.n, 256, 1, s, (double)p.P * (4.0 * 256 + 32.0));
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("head_fwd");
}

template <int NT>
static void launch_strip_bwd(const StripBwd& p, long per, cnr_stream s) {
  static DeviceOnce attr_once;
  const size_t lds = (size_t)8 * NT * 256 * sizeof(float);
  if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&strip_bwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((strip_bwd_kernel<NT>), dim3(p.nslots), dim3(512), lds, s, p, per);
}
void be_strip_bwd(const StripBwd& p, cnr_stream s) {
  const long per = round_up((int)((p.P + p.nslots - 1) / p.nslots), 64);
  TimingScope ts_("strip_bwd", 2, p.nt, p.P, 256, p.nt, 1, s, (double)p.P * (4.0 * 256 + 8.0 * p.nt));
  if (p.nt <= 4) launch_strip_bwd<4>(p, per, s); else launch_strip_bwd<8>(p, per, s);
  CNR_LAUNCH_CHECK("strip_bwd");
}

// ================================================================================================
// point-wise kernels
// ================================================================================================
#define CNR_PW_KERNEL(NAME, PARAM, BODY)                                                    \
  __global__ __launch_bounds__(256) void NAME##_kernel(const PARAM p, long n) {              \
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)   \
      BODY(p, i);                                                                            \
  }                                                                                          \
  static void NAME##_launch(const PARAM& p, long n, cnr_stream s) {                          \
    if (n <= 0) return;                                                                      \
    long blocks = (n + 255) / 256;                                                           \
    if (blocks > 8192) blocks = 8192;                                                        \
    TimingScope ts_(#NAME, 2, 0, n, 0, 0, 0, s);                                             \
    hipLaunchKernelGGL(NAME##_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n);        \
    CNR_LAUNCH_CHECK(#NAME);                                                                 \
  }

// Point-wise kernels that write kEmb / kAux-wide ROWS (192 B per point): a thread computes its point's rows into LDS (row stride 49 floats:
// conflict-free), then the block stores the 128 rows -- one contiguous 24 KB piece of the output -- with fully coalesced accesses (a thread
// storing its own row touches 64 different rows per wave instruction: 1.4 TB/s against 4+ TB/s).
constexpr int kRowsPerBlock = 128, kRowLd = 49;
static_assert(kEmb == 48 && kAux == 48, "row staging assumes 48-float rows");
template <class PARAM, void (*ROWS)(const PARAM&, long, float*, float*)>
__global__ __launch_bounds__(kRowsPerBlock) void staged_rows_kernel(const PARAM p, long n, float* E, float* AUX) {
  __shared__ float sE[kRowsPerBlock * kRowLd];
  __shared__ float sA[kRowsPerBlock * kRowLd];
  const int tid = threadIdx.x;
  for (long base = (long)blockIdx.x * kRowsPerBlock; base < n; base += (long)gridDim.x * kRowsPerBlock) {
    if (base + tid < n) ROWS(p, base + tid, sE + tid * kRowLd, sA + tid * kRowLd);
    __syncthreads();
    const int cnt = (int)((n - base) < kRowsPerBlock ? (n - base) : kRowsPerBlock) * 48;
    for (int e = tid; e < cnt; e += kRowsPerBlock) {
      const int r = e / 48, c = e - r * 48;
      E[base * 48 + e] = sE[r * kRowLd + c];
      if (AUX) AUX[base * 48 + e] = sA[r * kRowLd + c];
    }
    __syncthreads();
  }
}
template <class PARAM, void (*ROWS)(const PARAM&, long, float*, float*)>
static void staged_rows_launch(const char* name, const PARAM& p, long n, float* E, float* AUX, cnr_stream s) {
  if (n <= 0) return;
  long blocks = (n + kRowsPerBlock - 1) / kRowsPerBlock;
  if (blocks > 16384) blocks = 16384;
  TimingScope ts_(name, 2, 0, n, 0, 0, 0, s);
  hipLaunchKernelGGL((staged_rows_kernel<PARAM, ROWS>), dim3((unsigned)blocks), dim3(kRowsPerBlock), 0, s, p, n, E, AUX);
}
static void embed_z_launch(const EmbedZ& p, long n, cnr_stream s) { staged_rows_launch<EmbedZ, body_embed_z_rows>("embed_z", p, n, p.E, nullptr, s); CNR_LAUNCH_CHECK("embed_z"); }
static void embed_pts_launch(const EmbedPts& p, long n, cnr_stream s) { staged_rows_launch<EmbedPts, body_embed_pts_rows>("embed_pts", p, n, p.E, p.AUX, s); CNR_LAUNCH_CHECK("embed_pts"); }
static void fine_setup_launch(const FineSetup& p, long n, cnr_stream s) { staged_rows_launch<FineSetup, body_fine_setup_rows>("fine_setup", p, n, p.E, p.AUX, s); CNR_LAUNCH_CHECK("fine_setup"); }
CNR_PW_KERNEL(coltop_bwd, ColTopBwd, body_coltop_bwd)
// N_OUTSIDE > 0 (NeRF++ background): plain per-ray / per-point kernels
CNR_PW_KERNEL(outside_z, OutsideZ, body_outside_z)
CNR_PW_KERNEL(outside_z_bwd, OutsideZBwd, body_outside_z_bwd)
CNR_PW_KERNEL(bg_embed, BgEmbed, body_bg_embed)
CNR_PW_KERNEL(bg_alpha, BgAlpha, body_bg_alpha)
CNR_PW_KERNEL(bg_heads_bwd, BgHeadsBwd, body_bg_heads_bwd)
CNR_PW_KERNEL(bg_join, BgJoin, body_bg_join)
CNR_PW_KERNEL(bg_embed_bwd, BgEmbedBwd, body_bg_embed_bwd)
CNR_PW_KERNEL(bg_rays_bwd, BgRaysBwd, body_bg_rays_bwd)
CNR_PW_KERNEL(composite_bg, CompositeBg, body_composite_bg)
CNR_PW_KERNEL(composite_bg_bwd, CompositeBgBwd, body_composite_bg_bwd)
void be_outside_z(const OutsideZ& p, cnr_stream s) { outside_z_launch(p, p.R, s); }
void be_outside_z_bwd(const OutsideZBwd& p, cnr_stream s) { outside_z_bwd_launch(p, p.R, s); }
void be_bg_embed(const BgEmbed& p, cnr_stream s) { bg_embed_launch(p, p.R * p.MF, s); }
void be_bg_alpha(const BgAlpha& p, cnr_stream s) { bg_alpha_launch(p, p.n, s); }
void be_bg_heads_bwd(const BgHeadsBwd& p, cnr_stream s) { bg_heads_bwd_launch(p, p.n, s); }
void be_bg_join(const BgJoin& p, cnr_stream s) { bg_join_launch(p, p.n * p.W, s); }
void be_bg_embed_bwd(const BgEmbedBwd& p, cnr_stream s) { bg_embed_bwd_launch(p, p.R * p.MF, s); }
void be_bg_rays_bwd(const BgRaysBwd& p, cnr_stream s) { bg_rays_bwd_launch(p, p.R, s); }
void be_composite_bg(const CompositeBg& p, cnr_stream s) { composite_bg_launch(p, p.R, s); }
void be_composite_bg_bwd(const CompositeBgBwd& p, cnr_stream s) { composite_bg_bwd_launch(p, p.f.R, s); }
CNR_PW_KERNEL(pbar_finish, PbarFinish, body_pbar_finish)
CNR_PW_KERNEL(gen_rays, GenRays, body_gen_rays)

void be_embed_z(const EmbedZ& p, cnr_stream s) { embed_z_launch(p, p.R * p.m, s); }
void be_embed_pts(const EmbedPts& p, cnr_stream s) { embed_pts_launch(p, p.n, s); }
void be_fine_setup(const FineSetup& p, cnr_stream s) { fine_setup_launch(p, p.R * p.M, s); }
void be_coltop_bwd(const ColTopBwd& p, cnr_stream s) { coltop_bwd_launch(p, p.P, s); }
void be_pbar_finish(const PbarFinish& p, cnr_stream s) { pbar_finish_launch(p, p.P, s); }


// ------------------------------------------------------------------------------------------------
// PE-Jacobian kernels: 16 lanes per point, lane j owns the column triple [3j, 3j+3) of the 48-wide rows
// (triple 0 = x, triple 1+2k = sin(2^k x), triple 2+2k = cos(2^k x)), so every row is read/written as one contiguous
// 192-byte segment by its 16 lanes; the sin/cos partner values travel by shuffle.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_finish_kernel(const GradFinish p) {
  const int lane16 = threadIdx.x & 15;
  const long pt0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const long stride = (long)gridDim.x * 16;
  for (long pt = pt0; pt < p.P; pt += stride) {   // all 16 lanes of a group share pt -> uniform trip count inside a group
    const int c0 = lane16 * 3;
    float e[3], ce[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e[c] = p.E[pt * kEmb + c0 + c];
      ce[c] = p.ce0[pt * kEmb + c0 + c] + (p.ces ? p.ces[pt * kEmb + c0 + c] : 0.0f);
    }
    const int k = lane16 > 0 ? (lane16 - 1) >> 1 : 0;
    const bool is_sin = lane16 >= 1 && (lane16 & 1) == 1 && lane16 <= 2 * p.multires;
    const bool is_cos = lane16 >= 2 && (lane16 & 1) == 0 && lane16 <= 2 * p.multires;
    const float f = (float)(1 << k);
    float part[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e_next = __shfl_down(e[c], 1, 16);   // sin lane reads cos(2^k x) from its right neighbour
      const float e_prev = __shfl_up(e[c], 1, 16);     // cos lane reads sin(2^k x) from its left neighbour
      float t = 0.0f;
      if (lane16 == 0) t = ce[c];
      else if (is_sin) t = f * (e_next * ce[c]);
      else if (is_cos) t = -(f * (e_prev * ce[c]));
      part[c] = t;
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1)
#pragma unroll
      for (int c = 0; c < 3; ++c) part[c] += __shfl_xor(part[c], d, 16);
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = part[c] * p.scale;
    if (lane16 == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { p.grad_out[pt * 3 + c] = g[c]; p.AUX[pt * kAux + 3 + c] = g[c]; }
    }
    float aux[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) aux[c] = lane16 == 1 ? g[c] : p.AUX[pt * kAux + c0 + c];   // row as it will read after the update
    if (p.neg_g_as_view) {   // AUX[6:] = PE(-g): triple t >= 2 holds encoding triple (t - 2)
      const int et = lane16 - 2;
      if (et >= 0) {
        float v[3] = {-g[0], -g[1], -g[2]};
        const int npe_tr = p.multires_view > 0 ? 1 + 2 * p.multires_view : 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float val = 0.0f;
          if (et == 0) val = v[c];
          else if (et < npe_tr) { const float ff = (float)(1 << ((et - 1) >> 1)); val = (et & 1) ? sinf(v[c] * ff) : cosf(v[c] * ff); }
          if (et < npe_tr) { aux[c] = val; p.AUX[pt * kAux + c0 + c] = val; }
        }
      }
    }
    if (p.featx) {
      const int w = p.ldfx - p.F;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (c0 + c < w) p.featx[pt * p.ldfx + p.F + c0 + c] = aux[c];
      for (int c = kAux + lane16; c < w; c += 16) p.featx[pt * p.ldfx + p.F + c] = 0.0f;
    }
  }
}
void be_grad_finish(const GradFinish& p, cnr_stream s) {
  if (p.P <= 0) return;
  long blocks = (p.P * 16 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  TimingScope ts_("grad_finish", 2, 0, p.P, 0, 0, 0, s);
  hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("grad_finish");
}

__global__ __launch_bounds__(256) void gbar_finish_kernel(const GbarFinish p) {
  const int lane16 = threadIdx.x & 15;
  const long pt0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const long stride = (long)gridDim.x * 16;
  for (long pt = pt0; pt < p.P; pt += stride) {
    const int c0 = lane16 * 3;
    float e[3], gb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e[c] = p.E[pt * kEmb + c0 + c];
      float v = p.gbar_alpha[pt * 4 + c];
      if (p.daux_c) v += p.daux_c[pt * kAux + 3 + c];
      if (p.daux_r) v += p.daux_r[pt * kAux + 3 + c];
      gb[c] = v;
    }
    const int k = lane16 > 0 ? (lane16 - 1) >> 1 : 0;
    const bool is_sin = lane16 >= 1 && (lane16 & 1) == 1 && lane16 <= 2 * p.multires;
    const bool is_cos = lane16 >= 2 && (lane16 & 1) == 0 && lane16 <= 2 * p.multires;
    const float f = (float)(1 << k);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e_next = __shfl_down(e[c], 1, 16);
      const float e_prev = __shfl_up(e[c], 1, 16);
      const float t = gb[c] * p.scale;
      float out = 0.0f;
      if (lane16 == 0) out = t;
      else if (is_sin) out = f * e_next * t;          // d g / d ce_sin = 2^k cos(2^k x0)
      else if (is_cos) out = -f * e_prev * t;         // d g / d ce_cos = -2^k sin(2^k x0)
      p.cbar[pt * kEmb + c0 + c] = out;
    }
    if (lane16 == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) p.gbar_total[pt * 4 + c] = gb[c];
      p.gbar_total[pt * 4 + 3] = 0.0f;
    }
  }
}
void be_gbar_finish(const GbarFinish& p, cnr_stream s) {
  if (p.P <= 0) return;
  long blocks = (p.P * 16 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  TimingScope ts_("gbar_finish", 2, 0, p.P, 0, 0, 0, s);
  hipLaunchKernelGGL(gbar_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("gbar_finish");
}

// ------------------------------------------------------------------------------------------------
// block reductions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x = fmaxf(x, __shfl_xor(x, d));
  return x;
}
__device__ __forceinline__ int wave_min_int(int x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { int y = __shfl_xor(x, d); x = y < x ? y : x; }
  return x;
}
// sum over a 256-thread block; result valid in every thread
__device__ __forceinline__ float block_sum_256(float x, float* sh4 /*>= 4 floats*/) {
  x = wave_sum(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = x;
  __syncthreads();
  return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

__device__ __forceinline__ double block_sum_256d(double x, double* sh4) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = x;
  __syncthreads();
  return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

// effective weights: weight-norm, column permutation, zero padding, transpose copy.  One block per padded row.
__device__ __forceinline__ void prep_weight_row(const PrepWeight& p, const int n, double* redd) {
  const int tid = threadIdx.x;                     // n: internal row
  const bool real = n < p.n;
  const int nr = real ? (n + p.row_rot) % p.n : 0;  // reference row
  float scale = 1.0f;
  if (p.g != nullptr) {
    double ss = 0.0;   // row norm in double: weight_norm is the most rounding-sensitive step (x inv_s downstream)
    if (real)
      for (int c = tid; c < p.k_ref; c += 256) { double x = p.v[(long)nr * p.k_ref + c]; ss += x * x; }
    ss = block_sum_256d(ss, redd);
    if (real) scale = p.g[nr] / (float)sqrt(ss);
  }
  for (int j = tid; j < p.kpad; j += 256) {
    float val = 0.0f;
    if (real && j < p.ldw) {
      int src = -1;
      for (int q = 0; q < p.nseg; ++q)
        if (j >= p.seg[q].dst && j < p.seg[q].dst + p.seg[q].len) src = p.seg[q].src + (j - p.seg[q].dst);
      if (src >= 0) val = p.v[(long)nr * p.k_ref + src] * scale;
    }
    if (j < p.ldw) p.W[(long)n * p.ldw + j] = val;
    if (n < p.ldwt) p.Wt[(long)j * p.ldwt + n] = val;
  }
  if (tid == 0) p.bias[n] = real && p.b ? p.b[nr] : 0.0f;
}
__global__ __launch_bounds__(256) void prep_weight_kernel(const PrepWeight p) {
  __shared__ double redd[4];
  prep_weight_row(p, blockIdx.x, redd);
}
constexpr int kPrepBatch = 24;
struct PrepBatch { int count; int row_start[kPrepBatch + 1]; PrepWeight p[kPrepBatch]; };
static_assert(sizeof(PrepBatch) <= 4096, "kernel argument block");
__global__ __launch_bounds__(256) void prep_weight_batch_kernel(const PrepBatch b) {
  __shared__ double redd[4];
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.row_start[i + 1]) ++i;
  prep_weight_row(b.p[i], (int)blockIdx.x - b.row_start[i], redd);
}

// fp32 matrix -> two f16 planes of the row-scaled matrix: x * 2^e = hi + lo with e chosen per row so that max|x| * 2^e is in
// [2^13, 2^14) (exact scaling; 11 + 11 significand bits), plus 1 / 2^e per row.  One 64-thread block per row.
__device__ __forceinline__ void split_planes_row(const float* src, int ld, unsigned short* planes, long plane_stride, float* inv_scale, const int r) {
  const float* row = src + (long)r * ld;
  float mx = 0.0f;
  for (int k = threadIdx.x; k < ld; k += 64) mx = fmaxf(mx, fabsf(row[k]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  float sc = 1.0f;
  if (mx > 0.0f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); if (e < -100) e = -100; sc = ldexpf(1.0f, 14 - e); }
  for (int k = threadIdx.x; k < ld; k += 64) {
    const float x = row[k] * sc;
    const _Float16 h1 = (_Float16)x;
    const _Float16 h2 = (_Float16)(x - (float)h1);
    planes[(long)r * ld + k] = __builtin_bit_cast(unsigned short, h1);
    planes[plane_stride + (long)r * ld + k] = __builtin_bit_cast(unsigned short, h2);
  }
  if (threadIdx.x == 0) inv_scale[r] = 1.0f / sc;
}
__global__ __launch_bounds__(64) void split_planes_kernel(const float* src, int ld, unsigned short* planes, long plane_stride, float* inv_scale) {
  split_planes_row(src, ld, planes, plane_stride, inv_scale, blockIdx.x);
}
constexpr int kSplitBatch = 2 * kPrepBatch;
struct SplitBatch { int count; int row_start[kSplitBatch + 1]; SplitJob j[kSplitBatch]; };
static_assert(sizeof(SplitBatch) <= 4096, "kernel argument block");
__global__ __launch_bounds__(64) void split_planes_batch_kernel(const SplitBatch b) {
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.row_start[i + 1]) ++i;
  const SplitJob& j = b.j[i];
  split_planes_row(j.src, j.ld, j.planes, (long)j.rows * j.ld, j.inv_scale, (int)blockIdx.x - b.row_start[i]);
}
void be_split_planes_many(const SplitJob* jobs, int count, cnr_stream s) {
  for (int i0 = 0; i0 < count; i0 += kSplitBatch) {
    SplitBatch b;
    b.count = count - i0 < kSplitBatch ? count - i0 : kSplitBatch;
    int rows = 0;
    for (int i = 0; i < b.count; ++i) { b.row_start[i] = rows; b.j[i] = jobs[i0 + i]; rows += jobs[i0 + i].rows; }
    b.row_start[b.count] = rows;
    TimingScope ts_("split_planes", 2, 0, rows, 0, 0, 0, s);
    hipLaunchKernelGGL(split_planes_batch_kernel, dim3(rows), dim3(64), 0, s, b);
  }
  CNR_LAUNCH_CHECK("split_planes");
}
void be_prep_weights(const PrepWeight* p, int count, cnr_stream s) {
  for (int i0 = 0; i0 < count; i0 += kPrepBatch) {
    PrepBatch b;
    b.count = count - i0 < kPrepBatch ? count - i0 : kPrepBatch;
    int rows = 0;
    for (int i = 0; i < b.count; ++i) { b.row_start[i] = rows; b.p[i] = p[i0 + i]; rows += p[i0 + i].npad; }
    b.row_start[b.count] = rows;
    TimingScope ts_("prep_weight", 2, 0, rows, 0, 0, 0, s);
    hipLaunchKernelGGL(prep_weight_batch_kernel, dim3(rows), dim3(256), 0, s, b);
  }
  CNR_LAUNCH_CHECK("prep_weight");
}
void be_split_planes(const float* src, int rows, int ld, unsigned short* planes, float* inv_scale, cnr_stream s) {
  TimingScope ts_("split_planes", 2, 0, rows, 0, 0, 0, s);
  hipLaunchKernelGGL(split_planes_kernel, dim3(rows), dim3(64), 0, s, src, ld, planes, (long)rows * ld, inv_scale);
  CNR_LAUNCH_CHECK("split_planes");
}

void be_prep_weight(const PrepWeight& p, cnr_stream s) {
  TimingScope ts_("prep_weight", 2, 0, p.npad, 0, 0, 0, s);
  hipLaunchKernelGGL(prep_weight_kernel, dim3(p.npad), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("prep_weight");
}

// reduce dW partials, undo the column permutation, weight-norm backward.  One block per output row.
__device__ __forceinline__ void finish_weight_row(const FinishWeight& p, const int n, float* red, double* redd, float* dwi) {
  const int tid = threadIdx.x;                        // n: internal row
  const int nr = (n + p.row_rot) % p.n;               // reference row
  for (int j = tid; j < p.ldk; j += 256) {
    // fixed-order reduction over the chunks, 8 independent loads in flight (same summation order as a plain loop)
    const float* src = p.partial + (long)n * p.ldk + j;
    const long cs = (long)p.npad * p.ldk;
    float s = 0.0f;
    int c = 0;
    const int nch = j >= p.col_hi ? p.nchunk_hi : p.nchunk;   // (strip columns: their own slot count, see StripBwd)
    for (; c + 8 <= nch; c += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[(long)(c + q) * cs];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; c < nch; ++c) s += src[(long)c * cs];
    dwi[j] = s;
  }
  __syncthreads();
  // gather into reference column order (held in registers: <= 2 columns per thread for k_ref <= 512)
  float dref[2] = {0.f, 0.f};
  int cref[2] = {-1, -1};
  int cnt = 0;
  for (int c = tid; c < p.k_ref && cnt < 2; c += 256, ++cnt) {
    int dst = -1;
    for (int q = 0; q < p.nseg; ++q)
      if (c >= p.seg[q].src && c < p.seg[q].src + p.seg[q].len) dst = p.seg[q].dst + (c - p.seg[q].src);
    cref[cnt] = c;
    dref[cnt] = dst >= 0 ? dwi[dst] : 0.0f;
  }
  if (p.g != nullptr) {
    double dotd = 0.0, ssd = 0.0;
    for (int q = 0; q < 2; ++q)
      if (cref[q] >= 0) { double vv = p.v[(long)nr * p.k_ref + cref[q]]; dotd += (double)dref[q] * vv; ssd += vv * vv; }
    dotd = block_sum_256d(dotd, redd);
    ssd = block_sum_256d(ssd, redd);
    const float dot = (float)dotd;
    const float nrm = (float)sqrt(ssd);
    const float gg = p.g[nr];
    if (tid == 0) p.dg[nr] = dot / nrm;
    for (int q = 0; q < 2; ++q)
      if (cref[q] >= 0) {
        float vv = p.v[(long)nr * p.k_ref + cref[q]];
        p.dv[(long)nr * p.k_ref + cref[q]] = (gg / nrm) * (dref[q] - dot / (nrm * nrm) * vv);
      }
  } else {
    for (int q = 0; q < 2; ++q)
      if (cref[q] >= 0) p.dv[(long)nr * p.k_ref + cref[q]] = dref[q];
  }
  if (p.db != nullptr && p.colsum != nullptr) {
    float s = 0.0f;
    const int ncs = p.ncolsum > 0 ? p.ncolsum : p.nchunk;
    for (int c = tid; c < ncs; c += 256) s += p.colsum[(long)c * p.npad + n];
    s = block_sum_256(s, red);
    if (tid == 0) p.db[nr] = s;
  }
}
__global__ __launch_bounds__(256) void finish_weight_kernel(const FinishWeight p) {
  __shared__ float red[4];
  __shared__ double redd[4];
  __shared__ float dwi[512];
  finish_weight_row(p, blockIdx.x, red, redd, dwi);
}
constexpr int kFinishBatch = 24;
struct FinishBatch { int count; int row_start[kFinishBatch + 1]; FinishWeight f[kFinishBatch]; };
static_assert(sizeof(FinishBatch) <= 4096, "kernel argument block");
__global__ __launch_bounds__(256) void finish_weight_batch_kernel(const FinishBatch b) {
  __shared__ float red[4];
  __shared__ double redd[4];
  __shared__ float dwi[512];
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.row_start[i + 1]) ++i;
  finish_weight_row(b.f[i], (int)blockIdx.x - b.row_start[i], red, redd, dwi);
}
void be_finish_weights(const FinishWeight* f, int count, cnr_stream s) {
  for (int i0 = 0; i0 < count; i0 += kFinishBatch) {
    FinishBatch b;
    b.count = count - i0 < kFinishBatch ? count - i0 : kFinishBatch;
    int rows = 0;
    for (int i = 0; i < b.count; ++i) { b.row_start[i] = rows; b.f[i] = f[i0 + i]; rows += f[i0 + i].n; }
    b.row_start[b.count] = rows;
    TimingScope ts_("finish_weight", 2, 0, rows, 0, 0, 0, s);
    hipLaunchKernelGGL(finish_weight_batch_kernel, dim3(rows), dim3(256), 0, s, b);
  }
  CNR_LAUNCH_CHECK("finish_weight");
}
void be_finish_weight(const FinishWeight& p, cnr_stream s) {
  TimingScope ts_("finish_weight", 2, 0, p.n, 0, 0, 0, s);
  hipLaunchKernelGGL(finish_weight_kernel, dim3(p.n), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("finish_weight");
}

// ================================================================================================
// training loss: block partial sums in a fixed order, then one block folds the partials; gradients are element-wise
// ================================================================================================
__global__ __launch_bounds__(256) void loss_partial_kernel(const LossArgs a, float* partial) {
  __shared__ float red[4];
  const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
  float s_rgb = 0.f, s_bce = 0.f, s_rel = 0.f;
  const long n_rgb = a.R * 3;
  for (long i = (long)b * 256 + tid; i < n_rgb; i += (long)nb * 256) s_rgb += loss_rgb_term(a.color[i], a.gt[i], a.rgb_l1);
  if (a.mask)
    for (long r = (long)b * 256 + tid; r < a.R; r += (long)nb * 256) s_bce += loss_bce_term(a.wsum[r], a.mask[r]);
  if (a.drel) {
    const long per_ray = a.drel_per_ray ? 1 : (long)a.M * 3, n_rel = a.R * per_ray;
    for (long i = (long)b * 256 + tid; i < n_rel; i += (long)nb * 256) {
      const float m = (a.include_mask && a.mask) ? a.mask[i / per_ray] : 1.0f;
      s_rel += a.drel[i] * m;
    }
  }
  s_rgb = block_sum_256(s_rgb, red);
  s_bce = block_sum_256(s_bce, red);
  s_rel = block_sum_256(s_rel, red);
  if (tid == 0) { partial[b * 4 + 0] = s_rgb; partial[b * 4 + 1] = s_bce; partial[b * 4 + 2] = s_rel; partial[b * 4 + 3] = 0.f; }
}
__global__ __launch_bounds__(256) void loss_fold_kernel(const float* partial, int nb, float* sums) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  for (int b = tid; b < nb; b += 256) { v[0] += partial[b * 4]; v[1] += partial[b * 4 + 1]; v[2] += partial[b * 4 + 2]; }
  for (int j = 0; j < 3; ++j) { const float s = block_sum_256(v[j], red); if (tid == 0) sums[j] = s; }
  if (tid == 0) sums[3] = 0.f;
}
void be_loss_sums(const LossArgs& a, float* partial, float* sums, cnr_stream s) {
  TimingScope ts_("loss_sums", 2, 0, a.R, 0, 0, 0, s);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(kLossBlocks), dim3(256), 0, s, a, partial);
  hipLaunchKernelGGL(loss_fold_kernel, dim3(1), dim3(256), 0, s, partial, kLossBlocks, sums);
  CNR_LAUNCH_CHECK("loss_sums");
}
// The whole forward side in ONE launch: every block writes its partial sums, takes a ticket, and the block that draws the last one folds all
// partials in the fixed order of loss_fold_kernel and runs the scalar tail (loss_combine).  Which block is last does not matter: bitwise the same
// sums as the two-launch form.  The ticket counter is a 4-byte slot of the CALLER's scratch (zero before the first use of the buffer; the last
// block leaves it zero): calls on one stream may share a buffer, calls that may overlap (other streams, other threads) bring their own.
// shard != 0 (ray-sharded runs): no scalar tail; the last block writes the rank's statistics for the all-reduce instead:
// stats[0..2] = the three sums, stats[3..4] = the rank's eikonal sums, stats[5] = a second copy of stats[4] that the all-reduce leaves alone.
__global__ __launch_bounds__(256) void loss_forward_kernel(const LossArgs a, float* partial, unsigned* ticket, const LossScalars c, const float* gerr, float* sums,
                                                           float* out6, const int shard, const float* eik_sums) {
  __shared__ float red[4];
  __shared__ int is_last;
  const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
  float s_rgb = 0.f, s_bce = 0.f, s_rel = 0.f;
  const long n_rgb = a.R * 3;
  for (long i = (long)b * 256 + tid; i < n_rgb; i += (long)nb * 256) s_rgb += loss_rgb_term(a.color[i], a.gt[i], a.rgb_l1);
  if (a.mask)
    for (long r = (long)b * 256 + tid; r < a.R; r += (long)nb * 256) s_bce += loss_bce_term(a.wsum[r], a.mask[r]);
  if (a.drel) {
    const long per_ray = a.drel_per_ray ? 1 : (long)a.M * 3, n_rel = a.R * per_ray;
    for (long i = (long)b * 256 + tid; i < n_rel; i += (long)nb * 256) {
      const float m = (a.include_mask && a.mask) ? a.mask[i / per_ray] : 1.0f;
      s_rel += a.drel[i] * m;
    }
  }
  s_rgb = block_sum_256(s_rgb, red);
  s_bce = block_sum_256(s_bce, red);
  s_rel = block_sum_256(s_rel, red);
  if (tid == 0) {
    partial[b * 4 + 0] = s_rgb; partial[b * 4 + 1] = s_bce; partial[b * 4 + 2] = s_rel; partial[b * 4 + 3] = 0.f;
    __threadfence();                                               // the partials are visible device-wide before the ticket is drawn
    const unsigned t = atomicAdd(ticket, 1u);
    is_last = t == (unsigned)nb - 1u;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  float v[3] = {0.f, 0.f, 0.f};
  for (int k = tid; k < nb; k += 256) {                            // (device-scope loads: other CUs wrote these)
    v[0] += __hip_atomic_load(partial + k * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[1] += __hip_atomic_load(partial + k * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[2] += __hip_atomic_load(partial + k * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __shared__ float tot[4];
  for (int j = 0; j < 3; ++j) { const float sj = block_sum_256(v[j], red); if (tid == 0) tot[j] = sj; }
  if (tid == 0) {
    tot[3] = 0.f;
    if (shard) {
      const float num = eik_sums[0], den = eik_sums[1];
      sums[0] = tot[0]; sums[1] = tot[1]; sums[2] = tot[2]; sums[3] = num; sums[4] = den; sums[5] = den; sums[6] = 0.f; sums[7] = 0.f;
    } else {
      sums[0] = tot[0]; sums[1] = tot[1]; sums[2] = tot[2]; sums[3] = 0.f;
      loss_combine(c, tot, gerr, out6);
    }
    *ticket = 0;
  }
}
void be_loss_forward(const LossArgs& a, float* partial, unsigned* ticket, const LossScalars& c, const float* gerr, float* sums, float* out6, cnr_stream s) {
  TimingScope ts_("loss_forward", 2, 0, a.R, 0, 0, 0, s);
  hipLaunchKernelGGL(loss_forward_kernel, dim3(kLossBlocks), dim3(256), 0, s, a, partial, ticket, c, gerr, sums, out6, 0, (const float*)nullptr);
  CNR_LAUNCH_CHECK("loss_forward");
}
void be_loss_shard_stats(const LossArgs& a, float* partial, unsigned* ticket, const float* eik_sums, float* stats8, cnr_stream s) {
  TimingScope ts_("loss_shard_stats", 2, 0, a.R, 0, 0, 0, s);
  hipLaunchKernelGGL(loss_forward_kernel, dim3(kLossBlocks), dim3(256), 0, s, a, partial, ticket, LossScalars{}, (const float*)nullptr, stats8, (float*)nullptr, 1, eik_sums);
  CNR_LAUNCH_CHECK("loss_shard_stats");
}
__global__ void loss_shard_combine_kernel(const LossScalars c, const float* stats, float* out8) { if (threadIdx.x == 0 && blockIdx.x == 0) loss_shard_combine(c, stats, out8); }
void be_loss_shard_combine(const LossScalars& c, const float* stats8, float* out8, cnr_stream s) {
  TimingScope ts_("loss_shard_combine", 2, 0, 1, 0, 0, 0, s);
  hipLaunchKernelGGL(loss_shard_combine_kernel, dim3(1), dim3(64), 0, s, c, stats8, out8);
  CNR_LAUNCH_CHECK("loss_shard_combine");
}
// ... and the backward side in one: every thread forms the coefficients from the upstream gradient itself (loss_coef: a dozen flops)
__global__ __launch_bounds__(256) void loss_backward_kernel(const LossArgs a, const LossScalars c, const float* g_loss, const float* mean_rel, const float* eik_factor,
                                                            float* coef_out, float* d_color, float* d_wsum, float* d_drel_ray) {
  float coef[4];
  loss_coef(c, g_loss, c.use_relight ? mean_rel : g_loss, eik_factor, coef);
  if (blockIdx.x == 0 && threadIdx.x == 0) { coef_out[0] = coef[0]; coef_out[1] = coef[1]; coef_out[2] = coef[2]; coef_out[3] = coef[3]; }
  const long n_rgb = a.R * 3;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_rgb; i += stride) d_color[i] = coef[0] * loss_rgb_grad(a.color[i], a.gt[i], a.rgb_l1);
  if (d_wsum)
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += stride) d_wsum[r] = a.mask ? coef[1] * loss_bce_grad(a.wsum[r], a.mask[r]) : 0.0f;
  if (d_drel_ray)   // d mean(delta_relight * mask)^2 / d delta_relight[r][j][c]: one value per ray
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += stride) d_drel_ray[r] = (a.include_mask && a.mask) ? coef[2] * a.mask[r] : coef[2];
}
void be_loss_backward(const LossArgs& a, const LossScalars& c, const float* g_loss, const float* mean_rel, const float* eik_factor, float* coef4, float* d_color,
                      float* d_wsum, float* d_drel_ray, cnr_stream s) {
  TimingScope ts_("loss_backward", 2, 0, a.R, 0, 0, 0, s);
  long blocks = (a.R * 3 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(loss_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, c, g_loss, mean_rel, eik_factor, coef4, d_color, d_wsum, d_drel_ray);
  CNR_LAUNCH_CHECK("loss_backward");
}
__global__ __launch_bounds__(256) void loss_grads_kernel(const LossArgs a, const float* coef, float* d_color, float* d_wsum, float* d_drel) {
  const float c_rgb = coef[0], c_bce = coef[1], c_rel = coef[2];
  const long n_rgb = a.R * 3, per_ray = (long)a.M * 3, n_rel = a.drel || d_drel ? a.R * per_ray : 0;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_rgb; i += stride) d_color[i] = c_rgb * loss_rgb_grad(a.color[i], a.gt[i], a.rgb_l1);
  if (d_wsum)
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += stride) d_wsum[r] = a.mask ? c_bce * loss_bce_grad(a.wsum[r], a.mask[r]) : 0.0f;
  if (d_drel)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_rel; i += stride) d_drel[i] = c_rel * ((a.include_mask && a.mask) ? a.mask[i / per_ray] : 1.0f);
}
__global__ void loss_combine_kernel(const LossScalars c, const float* sums, const float* gerr, float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) loss_combine(c, sums, gerr, out); }
__global__ void loss_coef_kernel(const LossScalars c, const float* g_loss, const float* mean_rel, float* coef) { if (threadIdx.x == 0 && blockIdx.x == 0) loss_coef(c, g_loss, mean_rel, nullptr, coef); }
void be_loss_combine(const LossScalars& c, const float* sums, const float* gerr, float* out6, cnr_stream s) {
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, s, c, sums, gerr, out6);
  CNR_LAUNCH_CHECK("loss_combine");
}
void be_loss_coef(const LossScalars& c, const float* g_loss, const float* mean_rel, float* coef4, cnr_stream s) {
  hipLaunchKernelGGL(loss_coef_kernel, dim3(1), dim3(64), 0, s, c, g_loss, mean_rel, coef4);
  CNR_LAUNCH_CHECK("loss_coef");
}
void be_loss_grads(const LossArgs& a, const float* coef, float* d_color, float* d_wsum, float* d_drel, cnr_stream s) {
  TimingScope ts_("loss_grads", 2, 0, a.R, 0, 0, 0, s);
  hipLaunchKernelGGL(loss_grads_kernel, dim3(1024), dim3(256), 0, s, a, coef, d_color, d_wsum, d_drel);
  CNR_LAUNCH_CHECK("loss_grads");
}

__global__ __launch_bounds__(256) void reduce_eik_kernel(const ReduceEik p) {
  __shared__ float red[4];
  float a = 0.0f, b = 0.0f;
  for (long r = threadIdx.x; r < p.R; r += 256) { a += p.partial[r * 2]; b += p.partial[r * 2 + 1]; }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) {
    p.sums[0] = a; p.sums[1] = b;
    if (p.sums_out) { p.sums_out[0] = a; p.sums_out[1] = b; }
    *p.gradient_error = a / (b + 1e-5f);
  }
}
void be_reduce_eik(const ReduceEik& p, cnr_stream s) {
  TimingScope ts_("reduce_eik", 2, 0, p.R, 0, 0, 0, s);
  hipLaunchKernelGGL(reduce_eik_kernel, dim3(1), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("reduce_eik");
}

__global__ __launch_bounds__(256) void variance_finish_kernel(const VarianceFinish p) {
  __shared__ float red[4];
  float a = 0.0f;
  for (long r = threadIdx.x; r < p.R; r += 256) a += p.partial[r];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) {
    float raw = expf(p.variance[0] * 10.0f);
    *p.d_variance = (raw >= 1e-6f && raw <= 1e6f) ? a * 10.0f * raw : 0.0f;
  }
}
void be_variance_finish(const VarianceFinish& p, cnr_stream s) {
  TimingScope ts_("variance_finish", 2, 0, p.R, 0, 0, 0, s);
  hipLaunchKernelGGL(variance_finish_kernel, dim3(1), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("variance_finish");
}

// ================================================================================================
// per-ray kernels: one wavefront per ray, 4 rays per workgroup, ray state staged in LDS
// ================================================================================================
__device__ __forceinline__ float wave_scan_incl_add(float x, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { float y = __shfl_up(x, d); if (lane >= d) x = y + x; }
  return x;
}
__device__ __forceinline__ float wave_scan_incl_mul(float x, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { float y = __shfl_up(x, d); if (lane >= d) x = y * x; }
  return x;
}
// reverse inclusive scan: result[lane] = sum_{l >= lane} x[l]
__device__ __forceinline__ float wave_scan_incl_add_rev(float x, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { float y = __shfl_down(x, d); if (lane + d < 64) x = x + y; }
  return x;
}

__global__ __launch_bounds__(256) void upsample_kernel(const UpSample p) {
  __shared__ float sh[4][5][kMaxRaySamples];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long ray = (long)blockIdx.x * 4 + wave;
  const bool active = ray < p.R;
  if (!active) ray = p.R - 1;
  float* zs = sh[wave][0]; float* ss = sh[wave][1]; float* cs = sh[wave][2]; float* rs = sh[wave][3]; float* as = sh[wave][4];
  const int n = p.n, nsec = p.n - 1;
  const bool given_w = p.w_in != nullptr;   // wave-uniform
  if (given_w) {
    for (int i = lane; i < n; i += 64) zs[i] = p.z[ray * p.ldz + i];
    for (int i = lane; i < nsec; i += 64) as[i] = p.w_in[ray * nsec + i];
  } else {
    float o[3], d[3];
    for (int c = 0; c < 3; ++c) { o[c] = p.o[ray * 3 + c]; d[c] = p.d[ray * 3 + c]; }
    for (int i = lane; i < n; i += 64) {
      float z = p.z[ray * p.ldz + i];
      zs[i] = z;
      ss[i] = p.sdf[ray * p.lds + i];
      float x = o[0] + d[0] * z, y = o[1] + d[1] * z, w = o[2] + d[2] * z;
      rs[i] = sqrtf(x * x + y * y + w * w);
    }
  }
  __syncthreads();
  if (!given_w) for (int i = lane; i < nsec; i += 64) cs[i] = (ss[i + 1] - ss[i]) / (zs[i + 1] - zs[i] + 1e-5f);
  __syncthreads();
  if (!given_w)
    for (int i = lane; i < nsec; i += 64) {
      float prev = i > 0 ? cs[i - 1] : 0.0f;
      float c = fminf(prev, cs[i]);
      c = fminf(fmaxf(c, -1e3f), 0.0f);
      const bool inside = rs[i] < 1.0f || rs[i + 1] < 1.0f;
      c = inside ? c : c * 0.0f;
      as[i] = upsample_alpha(ss[i], ss[i + 1], zs[i], zs[i + 1], c, p.inv_s);
    }
  __syncthreads();
  // weights = alpha * exclusive cumprod(1 - alpha + 1e-7), + 1e-5 (sample_pdf), and their sum
  float carry = 1.0f, part = 0.0f;
  for (int base = 0; base < nsec; base += 64) {
    const int i = base + lane;
    const bool ok = i < nsec;
    const float a = ok ? as[i] : 0.0f;
    const float f = ok ? 1.0f - a + 1e-7f : 1.0f;
    const float incl = wave_scan_incl_mul(f, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0f;
    const float w = (given_w ? a : a * (carry * excl)) + 1e-5f;
    if (ok) { as[i] = w; part += w; }
    carry = carry * __shfl(incl, 63);
  }
  const float total = wave_sum(part);
  __syncthreads();
  // cdf (n entries, cdf[0] = 0) -> cs
  float csum = 0.0f;
  for (int base = 0; base < nsec; base += 64) {
    const int i = base + lane;
    const bool ok = i < nsec;
    const float pdf = ok ? as[i] / total : 0.0f;
    const float incl = wave_scan_incl_add(pdf, lane);
    if (ok) cs[i + 1] = csum + incl;
    csum = csum + __shfl(incl, 63);
  }
  if (lane == 0) cs[0] = 0.0f;
  __syncthreads();
  if (lane < p.m && active) {
    const float u = p.u_in ? p.u_in[ray * p.m + lane] : linspace_at(0.5f / (float)p.m, 1.0f - 0.5f / (float)p.m, p.m, lane);
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cs[mid] > u) hi = mid; else lo = mid + 1; }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < n - 1 ? lo : n - 1;
    const float c0 = cs[below], c1 = cs[above];
    const float b0 = zs[below], b1 = zs[above];
    float den = c1 - c0;
    if (den < 1e-5f) den = 1.0f;
    const float t = (u - c0) / den;
    p.new_z[ray * p.m + lane] = b0 + t * (b1 - b0);
  }
}
void be_upsample(const UpSample& p, cnr_stream s) {
  // algorithmic bytes per ray: z and sdf (or the given weights) in, o / d, the new positions out
  TimingScope ts_("upsample", 2, 0, p.R, 0, 0, 0, s, (double)p.R * (4.0 * p.n + 4.0 * (p.w_in ? p.n - 1 : p.n) + (p.w_in ? 0.0 : 24.0) + 4.0 * p.m));
  hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("upsample");
}

__global__ __launch_bounds__(256) void merge_kernel(const MergeZ p) {
  __shared__ float sh[4][2][kMaxRaySamples];
  __shared__ float shn[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long ray = (long)blockIdx.x * 4 + wave;
  const bool active = ray < p.R;
  if (!active) ray = p.R - 1;
  float* zs = sh[wave][0]; float* ss = sh[wave][1]; float* nz = shn[wave][0]; float* ns = shn[wave][1];
  const bool with_sdf = p.new_sdf != nullptr;
  for (int i = lane; i < p.n; i += 64) { zs[i] = p.z[ray * p.ldz + i]; ss[i] = with_sdf ? p.sdf_in[ray * p.lds_in + i] : 0.0f; }
  for (int j = lane; j < p.m; j += 64) { nz[j] = p.new_z[ray * p.m + j]; ns[j] = with_sdf ? p.new_sdf[ray * p.m + j] : 0.0f; }
  __syncthreads();
  if (active) {
    for (int i = lane; i < p.n; i += 64) {   // old sample i: stable position = i + #(new < z_i)
      const float z = zs[i];
      int cnt = 0;
      for (int j = 0; j < p.m; ++j) cnt += nz[j] < z ? 1 : 0;
      p.z[ray * p.ldz + i + cnt] = z;
      if (with_sdf) p.sdf_out[ray * p.lds_out + i + cnt] = ss[i];
    }
    for (int j = lane; j < p.m; j += 64) {   // new sample j: position = j + #(old <= new_j), new samples keep their order
      const float z = nz[j];
      int lo = 0, hi = p.n;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (zs[mid] > z) hi = mid; else lo = mid + 1; }
      int rank = 0;   // rank among the new samples (they are monotone in practice; this keeps the merge a bijection regardless)
      for (int q2 = 0; q2 < p.m; ++q2) rank += (nz[q2] < z || (nz[q2] == z && q2 < j)) ? 1 : 0;
      p.z[ray * p.ldz + lo + rank] = z;
      if (with_sdf) p.sdf_out[ray * p.lds_out + lo + rank] = ns[j];
    }
  }
}
// merge (previous iteration) + up_sample + embedding of the new samples for one ray per wavefront: the arithmetic of merge_kernel,
// upsample_kernel and the EmbedZ body, with the merged row handed over in LDS instead of through three launches
__global__ __launch_bounds__(256) void sampler_step_kernel(const SamplerStep p) {
  __shared__ float sh[4][7][kMaxRaySamples];
  __shared__ float shn[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long ray = (long)blockIdx.x * 4 + wave;
  const bool active = ray < p.u.R;
  if (!active) ray = p.u.R - 1;
  float* zs = sh[wave][0]; float* ss = sh[wave][1]; float* cs = sh[wave][2]; float* rs = sh[wave][3]; float* as = sh[wave][4];
  float* zo = sh[wave][5]; float* so = sh[wave][6]; float* nz = shn[wave][0]; float* ns = shn[wave][1];
  const UpSample& u = p.u;
  const int n = u.n, nsec = u.n - 1;
  float o[3], d[3];
  for (int c = 0; c < 3; ++c) { o[c] = u.o[ray * 3 + c]; d[c] = u.d[ray * 3 + c]; }
  if (p.do_merge) {
    const MergeZ& g = p.g;   // (with new_sdf: the fused form is only used between two up-sampling iterations)
    for (int i = lane; i < g.n; i += 64) { zo[i] = g.z[ray * g.ldz + i]; so[i] = g.sdf_in[ray * g.lds_in + i]; }
    for (int j = lane; j < g.m; j += 64) { nz[j] = g.new_z[ray * g.m + j]; ns[j] = g.new_sdf[ray * g.m + j]; }
    __syncthreads();
    for (int i = lane; i < g.n; i += 64) {   // old sample i: stable position = i + #(new < z_i)
      const float z = zo[i];
      int cnt = 0;
      for (int j = 0; j < g.m; ++j) cnt += nz[j] < z ? 1 : 0;
      zs[i + cnt] = z; ss[i + cnt] = so[i];
    }
    for (int j = lane; j < g.m; j += 64) {   // new sample j: position = j + #(old <= new_j), new samples keep their order
      const float z = nz[j];
      int lo = 0, hi = g.n;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (zo[mid] > z) hi = mid; else lo = mid + 1; }
      int rank = 0;
      for (int q2 = 0; q2 < g.m; ++q2) rank += (nz[q2] < z || (nz[q2] == z && q2 < j)) ? 1 : 0;
      zs[lo + rank] = z; ss[lo + rank] = ns[j];
    }
    __syncthreads();
    if (active)
      for (int i = lane; i < n; i += 64) { g.z[ray * g.ldz + i] = zs[i]; g.sdf_out[ray * g.lds_out + i] = ss[i]; }
  } else {
    for (int i = lane; i < n; i += 64) { zs[i] = u.z[ray * u.ldz + i]; ss[i] = u.sdf[ray * u.lds + i]; }
    __syncthreads();
  }
  for (int i = lane; i < n; i += 64) {
    const float z = zs[i];
    const float x = o[0] + d[0] * z, y = o[1] + d[1] * z, w = o[2] + d[2] * z;
    rs[i] = sqrtf(x * x + y * y + w * w);
  }
  __syncthreads();
  for (int i = lane; i < nsec; i += 64) cs[i] = (ss[i + 1] - ss[i]) / (zs[i + 1] - zs[i] + 1e-5f);
  __syncthreads();
  for (int i = lane; i < nsec; i += 64) {
    float prev = i > 0 ? cs[i - 1] : 0.0f;
    float c = fminf(prev, cs[i]);
    c = fminf(fmaxf(c, -1e3f), 0.0f);
    const bool inside = rs[i] < 1.0f || rs[i + 1] < 1.0f;
    c = inside ? c : c * 0.0f;
    as[i] = upsample_alpha(ss[i], ss[i + 1], zs[i], zs[i + 1], c, u.inv_s);
  }
  __syncthreads();
  float carry = 1.0f, part = 0.0f;
  for (int base = 0; base < nsec; base += 64) {
    const int i = base + lane;
    const bool ok = i < nsec;
    const float a = ok ? as[i] : 0.0f;
    const float f = ok ? 1.0f - a + 1e-7f : 1.0f;
    const float incl = wave_scan_incl_mul(f, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0f;
    const float w = a * (carry * excl) + 1e-5f;
    if (ok) { as[i] = w; part += w; }
    carry = carry * __shfl(incl, 63);
  }
  const float total = wave_sum(part);
  __syncthreads();
  float csum = 0.0f;
  for (int base = 0; base < nsec; base += 64) {
    const int i = base + lane;
    const bool ok = i < nsec;
    const float pdf = ok ? as[i] / total : 0.0f;
    const float incl = wave_scan_incl_add(pdf, lane);
    if (ok) cs[i + 1] = csum + incl;
    csum = csum + __shfl(incl, 63);
  }
  if (lane == 0) cs[0] = 0.0f;
  __syncthreads();
  if (lane < u.m) {
    const float uu = linspace_at(0.5f / (float)u.m, 1.0f - 0.5f / (float)u.m, u.m, lane);
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cs[mid] > uu) hi = mid; else lo = mid + 1; }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < n - 1 ? lo : n - 1;
    const float c0 = cs[below], c1 = cs[above];
    const float b0 = zs[below], b1 = zs[above];
    float den = c1 - c0;
    if (den < 1e-5f) den = 1.0f;
    const float t = (uu - c0) / den;
    const float znew = b0 + t * (b1 - b0);
    nz[lane] = znew;
    if (active) u.new_z[ray * u.m + lane] = znew;
  }
  if (p.do_embed) {
    __syncthreads();
    if (active) {
      // triples of the kEmb-wide rows of the u.m new samples (one contiguous piece of E): triple 0 = x, 1 + 2k = sin(2^k x), 2 + 2k = cos(2^k x)
      const int ntr = kEmb / 3;
      for (int idx = lane; idx < u.m * ntr; idx += 64) {
        const int j = idx / ntr, t = idx - j * ntr;
        const float z = nz[j];
        float v[3];
        const int k = t > 0 ? (t - 1) >> 1 : 0;
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float x = (o[c] + d[c] * z) * p.scale;
          float sn, cn;
          sincosf(x * f, &sn, &cn);
          v[c] = t == 0 ? x : (t <= 2 * p.multires ? ((t & 1) ? sn : cn) : 0.0f);
        }
        float* e = p.E + (ray * u.m + j) * kEmb + 3 * t;
        e[0] = v[0]; e[1] = v[1]; e[2] = v[2];
      }
    }
  }
}
void be_sampler_step(const SamplerStep& p, cnr_stream s) {
  TimingScope ts_("sampler_step", 2, 0, p.u.R, 0, 0, 0, s, (double)p.u.R * (8.0 * p.u.n + 24.0 + 4.0 * p.u.m + (p.do_embed ? 4.0 * kEmb * p.u.m : 0.0)));
  hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)((p.u.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("sampler_step");
}

void be_merge(const MergeZ& p, cnr_stream s) {
  TimingScope ts_("merge", 2, 0, p.R, 0, 0, 0, s, (double)p.R * (p.new_sdf ? 2.0 : 1.0) * (4.0 * p.n + 4.0 * p.m + 4.0 * (p.n + p.m)));
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("merge");
}

constexpr int kRayChunks = kMaxRaySamples / 64;

struct RaySample {   // forward quantities of one sample, recomputed identically in forward and backward
  bool ok;
  float z, dist, relax, inside, gn;
  float g[3];
  AlphaOut a;
};

__device__ __forceinline__ RaySample ray_sample(const float* zs, int j, int M, float sample_dist, const float o[3], const float d[3],
                                                const float* sdf, const float* g, long pt, float inv_s, float r) {
  RaySample q;
  q.ok = j < M;
  const int jj = q.ok ? j : M - 1;
  q.z = zs[jj];
  q.dist = jj + 1 < M ? zs[jj + 1] - q.z : sample_dist;
  const float mid = q.z + q.dist * 0.5f;
  const float x = o[0] + d[0] * mid, y = o[1] + d[1] * mid, w = o[2] + d[2] * mid;
  const float pn = sqrtf(x * x + y * y + w * w);
  q.inside = pn < 1.0f ? 1.0f : 0.0f;
  q.relax = pn < 1.2f ? 1.0f : 0.0f;
  const long p2 = q.ok ? pt : pt - (j - jj);
  q.g[0] = g[p2 * 3]; q.g[1] = g[p2 * 3 + 1]; q.g[2] = g[p2 * 3 + 2];
  q.gn = sqrtf(q.g[0] * q.g[0] + q.g[1] * q.g[1] + q.g[2] * q.g[2]);
  q.a = alpha_forward(sdf[p2], q.g, d, q.dist, inv_s, r);
  return q;
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(const CompositeFwd p) {
  __shared__ float shz[4][kMaxRaySamples];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long ray = (long)blockIdx.x * 4 + wave;
  const bool active = ray < p.R;
  if (!active) ray = p.R - 1;
  float* zs = shz[wave];
  const int M = p.M;
  for (int i = lane; i < M; i += 64) zs[i] = p.z[ray * M + i];
  __syncthreads();
  float o[3], d[3];
  for (int c = 0; c < 3; ++c) { o[c] = p.o[ray * 3 + c]; d[c] = p.d[ray * 3 + c]; }
  const float inv_s = fminf(fmaxf(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);

  float carry = 1.0f;
  float wsum = 0.f, wmax = -1.f, dep = 0.f, col[3] = {0.f, 0.f, 0.f}, gcl[3] = {0.f, 0.f, 0.f}, e0 = 0.f, e1 = 0.f, drs = 0.f;
#pragma unroll
  for (int c = 0; c < kRayChunks; ++c) {
    if (c * 64 < M) {
      const int j = c * 64 + lane;
      const long pt = ray * M + j;
      if (p.delta && j < M) drs += (p.delta[pt * 3] + p.delta[pt * 3 + 1]) + p.delta[pt * 3 + 2];
      RaySample q = ray_sample(zs, j, M, p.sample_dist, o, d, p.sdf, p.g, pt, inv_s, p.cos_anneal);
      const float f = q.ok ? 1.0f - q.a.alpha + 1e-7f : 1.0f;
      const float incl = wave_scan_incl_mul(f, lane);
      float excl = __shfl_up(incl, 1);
      if (lane == 0) excl = 1.0f;
      const float w = q.ok ? q.a.alpha * (carry * excl) : 0.0f;
      carry = carry * __shfl(incl, 63);
      if (q.ok) {
        wsum += w; wmax = fmaxf(wmax, w); dep += w * q.z;
        if (p.color) for (int k = 0; k < 3; ++k) col[k] += w * p.color[pt * p.ldcolor + k];   // (null: the weights-only pass in front of the compaction)
        if (p.gcolor) for (int k = 0; k < 3; ++k) gcl[k] += w * p.gcolor[pt * p.ldg + k];
        e0 += q.relax * (q.gn - 1.0f) * (q.gn - 1.0f);
        e1 += q.relax;
        if (active) {
          p.weights[pt] = w;
          p.cdf_fine[pt] = q.a.pc;
          p.inside_sphere[pt] = q.inside;
          if (p.sdf_s) p.sdf_s[pt] = p.sdf[pt];
          if (p.color_s && p.color) for (int k = 0; k < 3; ++k) p.color_s[pt * 3 + k] = p.color[pt * p.ldcolor + k];
          if (p.gcolor_s && p.gcolor) for (int k = 0; k < 3; ++k) p.gcolor_s[pt * 3 + k] = p.gcolor[pt * p.ldg + k];
        }
      }
    }
  }
  wsum = wave_sum(wsum); wmax = wave_max(wmax); dep = wave_sum(dep);
  for (int k = 0; k < 3; ++k) { col[k] = wave_sum(col[k]); gcl[k] = wave_sum(gcl[k]); }
  e0 = wave_sum(e0); e1 = wave_sum(e1);
  if (p.delta_ray_sum) { drs = wave_sum(drs); if (lane == 0 && active) p.delta_ray_sum[ray] = drs; }
  if (lane == 0 && active) {
    for (int k = 0; k < 3; ++k) {
      float cc = col[k];
      if (p.background_rgb) cc = cc + p.background_rgb[k] * (1.0f - wsum);
      p.color_fine[ray * 3 + k] = cc;
      if (p.global_color) p.global_color[ray * 3 + k] = gcl[k];
    }
    p.weight_sum[ray] = wsum; p.weight_max[ray] = wmax; p.depth[ray] = dep;
    p.s_val[ray] = 1.0f / inv_s;
    p.eik_partial[ray * 2] = e0; p.eik_partial[ray * 2 + 1] = e1;
  }
}
void be_composite_fwd(const CompositeFwd& p, cnr_stream s) {
  // algorithmic bytes per ray: z, sdf, gradients, colour(s) in; cdf, weights, inside_sphere and the per-ray outputs out
  TimingScope ts_("composite_fwd", 2, 0, p.R, 0, 0, 0, s,
                  (double)p.R * ((4.0 + 4.0 + 12.0 + 12.0 + (p.gcolor ? 12.0 : 0.0)) * p.M + 24.0 + 12.0 * p.M + 52.0));
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("composite_fwd");
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(const CompositeBwd p) {
  __shared__ float shz[4][kMaxRaySamples];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long ray = (long)blockIdx.x * 4 + wave;
  const bool active = ray < p.R;
  if (!active) ray = p.R - 1;
  float* zs = shz[wave];
  const int M = p.M;
  for (int i = lane; i < M; i += 64) zs[i] = p.z[ray * M + i];
  __syncthreads();
  float o[3], d[3];
  for (int c = 0; c < 3; ++c) { o[c] = p.o[ray * 3 + c]; d[c] = p.d[ray * 3 + c]; }
  const float inv_s = fminf(fmaxf(expf(p.variance[0] * 10.0f), 1e-6f), 1e6f);

  RaySample q[kRayChunks];
  float T[kRayChunks], w[kRayChunks];
  float carry = 1.0f, wsum = 0.0f, wmax = -1.0f;
#pragma unroll
  for (int c = 0; c < kRayChunks; ++c) {
    T[c] = 0.f; w[c] = 0.f;
    if (c * 64 < M) {
      const int j = c * 64 + lane;
      q[c] = ray_sample(zs, j, M, p.sample_dist, o, d, p.sdf, p.g, ray * M + j, inv_s, p.cos_anneal);
      const float f = q[c].ok ? 1.0f - q[c].a.alpha + 1e-7f : 1.0f;
      const float incl = wave_scan_incl_mul(f, lane);
      float excl = __shfl_up(incl, 1);
      if (lane == 0) excl = 1.0f;
      T[c] = carry * excl;
      w[c] = q[c].ok ? q[c].a.alpha * T[c] : 0.0f;
      carry = carry * __shfl(incl, 63);
      wsum += w[c];
      if (q[c].ok) wmax = fmaxf(wmax, w[c]);
    }
  }
  wsum = wave_sum(wsum);
  wmax = wave_max(wmax);
  int amax = 1 << 30;
#pragma unroll
  for (int c = 0; c < kRayChunks; ++c)
    if (c * 64 < M && q[c].ok && w[c] == wmax && c * 64 + lane < amax) amax = c * 64 + lane;
  amax = wave_min_int(amax);

  float dcol[3] = {0.f, 0.f, 0.f}, dglob[3] = {0.f, 0.f, 0.f};
  if (p.d_color_fine) for (int k = 0; k < 3; ++k) dcol[k] = p.d_color_fine[ray * 3 + k];
  if (p.d_global_color) for (int k = 0; k < 3; ++k) dglob[k] = p.d_global_color[ray * 3 + k];
  float dws = p.d_weight_sum ? p.d_weight_sum[ray] : 0.0f;
  if (p.background_rgb) for (int k = 0; k < 3; ++k) dws -= dcol[k] * p.background_rgb[k];
  const float ddepth = p.d_depth ? p.d_depth[ray] : 0.0f;
  const float dwmax = p.d_weight_max ? p.d_weight_max[ray] : 0.0f;
  const float dge = p.d_gradient_error ? p.d_gradient_error[0] : 0.0f;
  const float eik_den = p.eik_sums[1] + 1e-5f;
  const float ddrel_ray = p.d_delta_relight_ray ? p.d_delta_relight_ray[ray] : 0.0f;

  // d loss / d w_j and the suffix sums S_j = sum_{k>j} wbar_k w_k (reverse scan, chunks from the back)
  float wbar[kRayChunks], S[kRayChunks];
  float rcarry = 0.0f;
#pragma unroll
  for (int c = kRayChunks - 1; c >= 0; --c) {
    wbar[c] = 0.f; S[c] = 0.f;
    if (c * 64 < M) {
      const int j = c * 64 + lane;
      const long pt = ray * M + j;
      float wb = 0.0f;
      if (q[c].ok) {
        for (int k = 0; k < 3; ++k) wb += dcol[k] * p.color[pt * p.ldcolor + k];
        if (p.gcolor) for (int k = 0; k < 3; ++k) wb += dglob[k] * p.gcolor[pt * p.ldg + k];
        wb += dws + ddepth * q[c].z;
        if (p.d_weights) wb += p.d_weights[pt];
        if (j == amax) wb += dwmax;
      }
      wbar[c] = wb;
      const float x = wb * w[c];
      const float incl = wave_scan_incl_add_rev(x, lane);
      S[c] = rcarry + (incl - x);
      rcarry = rcarry + __shfl(incl, 0);
    }
  }

  float dinvs = 0.0f, drd[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < kRayChunks; ++c) {
    if (c * 64 < M) {
      const int j = c * 64 + lane;
      const long pt = ray * M + j;
      if (q[c].ok) {
        const float dalpha = wbar[c] * T[c] - S[c] / (1.0f - q[c].a.alpha + 1e-7f);
        const AlphaGrad ag = alpha_backward(q[c].a, q[c].dist, inv_s, p.cos_anneal, dalpha, p.d_cdf ? p.d_cdf[pt] : 0.0f);
        dinvs += ag.d_inv_s;
        float gb[3];
        const float ecoef = (q[c].relax > 0.0f && q[c].gn > 0.0f) ? dge / eik_den * 2.0f * (q[c].gn - 1.0f) / q[c].gn : 0.0f;
        for (int k = 0; k < 3; ++k) {
          gb[k] = ag.d_tc * d[k] + ecoef * q[c].g[k];
          if (p.d_gradients) gb[k] += p.d_gradients[pt * 3 + k];
          drd[k] += ag.d_tc * q[c].g[k];
        }
        if (active) {
          if (p.d_z) { p.d_z[pt * 2] = ddepth * w[c]; p.d_z[pt * 2 + 1] = ag.d_dist; }
          p.ztop[pt * p.ldztop + p.ztop_col] = (ag.d_sdf + (p.d_sdf_s ? p.d_sdf_s[pt] : 0.0f)) / p.sdf_scale;
          for (int k = p.ztop_col + 1; k < p.ldztop; ++k) p.ztop[pt * p.ldztop + k] = 0.0f;   // (the pad columns behind it: one launch less than zeroing them apart)
          for (int k = 0; k < 3; ++k) p.gbar[pt * 4 + k] = gb[k];
          p.gbar[pt * 4 + 3] = 0.0f;
          for (int k = 0; k < 3; ++k) {
            const float cbar = dcol[k] * w[c] + (p.d_color_s ? p.d_color_s[pt * 3 + k] : 0.0f);   // cotangent of the composited (relit) colour sample
            if (p.has_relight) {
              const float relit = p.color[pt * p.ldcolor + k];
              const float gc = p.gcolor[pt * p.ldg + k];
              float tbar, gca = dglob[k] * w[c] + (p.d_gcolor_s ? p.d_gcolor_s[pt * 3 + k] : 0.0f);
              if (p.inv_sigmoid) {
                tbar = cbar * relit * (1.0f - relit);
                gca += tbar * inverse_sigmoid_grad(gc);
              } else {
                const float pass = (relit > 0.0f && relit < 1.0f) ? 1.0f : 0.0f;   // clamp(rgb + sigmoid(h) - 0.5, 0, 1)
                const float sg = relit - gc + 0.5f;                                 // = sigmoid(h) where the clamp is inactive
                tbar = cbar * pass * sg * (1.0f - sg);
                gca += cbar * pass;
              }
              p.dtop[pt * p.ldtop + k] = tbar + (p.d_delta_relight ? p.d_delta_relight[pt * 3 + k] : 0.0f) + ddrel_ray;
              p.gc_a[pt * p.ldtop + k] = gca;
            } else {
              p.gc_a[pt * p.ldtop + k] = cbar;
            }
          }
          if (p.has_relight) for (int k = 3; k < p.ldtop; ++k) p.dtop[pt * p.ldtop + k] = 0.0f;
          for (int k = 3; k < p.ldtop; ++k) p.gc_a[pt * p.ldtop + k] = 0.0f;
        }
      }
    }
  }
  dinvs = wave_sum(dinvs);
  for (int k = 0; k < 3; ++k) drd[k] = wave_sum(drd[k]);
  if (lane == 0 && active) {
    if (p.d_s_val) dinvs += -p.d_s_val[ray] / (inv_s * inv_s);
    p.dinvs_partial[ray] = dinvs;
    if (p.d_rays_d) for (int k = 0; k < 3; ++k) p.d_rays_d[ray * 3 + k] = drd[k];
  }
}
void be_composite_bwd(const CompositeBwd& p, cnr_stream s) {
  // algorithmic bytes per ray: the forward inputs again, the upstream gradients, and per sample d sdf (4), d g (12), the colour /
  // relight cotangents (12 + 12) -- the buffers themselves are padded to the 16-float rows the narrow GEMMs read
  TimingScope ts_("composite_bwd", 2, 0, p.R, 0, 0, 0, s,
                  (double)p.R * ((4.0 + 4.0 + 12.0 + 12.0 + (p.gcolor ? 12.0 : 0.0)) * p.M + 24.0 + (p.d_delta_relight ? 12.0 * p.M : 0.0) + 32.0 +
                                 (4.0 + 12.0 + 12.0 + (p.has_relight ? 12.0 : 0.0)) * p.M + 16.0));
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("composite_bwd");
}


// ------------------------------------------------------------------------------------------------
// early-termination compaction (inference): wavefront ballot + popcount prefix per ray
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prune_count_kernel(const PruneCount p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= p.R) return;
  int cnt = 0;
  for (int base = 0; base < p.M; base += 64) {
    const int j = base + lane;
    const bool keep = j < p.M && p.weights[ray * p.M + j] >= p.eps;
    cnt += __popcll(__ballot(keep));
  }
  if (lane == 0) p.counts[ray] = cnt;
}
void be_prune_count(const PruneCount& p, cnr_stream s) {
  TimingScope ts_("prune_count", 2, 0, p.R, 0, 0, 0, s);
  hipLaunchKernelGGL(prune_count_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("prune_count");
}

// exclusive scan of the per-ray counts (one workgroup; R is a few thousand)
__global__ __launch_bounds__(1024) void prune_scan_kernel(const PruneScan p) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (long base = 0; base < p.R; base += 1024) {
    const long i = base + tid;
    int x = i < p.R ? p.counts[i] : 0;
    int incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (i < p.R) p.offsets[i] = carry + woff + incl - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) p.offsets[p.R] = carry_s;
}
void be_prune_scan(const PruneScan& p, cnr_stream s) {
  TimingScope ts_("prune_scan", 2, 0, p.R, 0, 0, 0, s);
  hipLaunchKernelGGL(prune_scan_kernel, dim3(1), dim3(1024), 0, s, p);
  CNR_LAUNCH_CHECK("prune_scan");
}

__global__ __launch_bounds__(256) void prune_gather_kernel(const PruneGather p) {
  __shared__ int list[4][kMaxRaySamples];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= p.R) return;
  int carry = 0;
  for (int base = 0; base < p.M; base += 64) {
    const int j = base + lane;
    const bool keep = j < p.M && p.weights[ray * p.M + j] >= p.eps;
    const unsigned long long mask = __ballot(keep);
    const int pos = carry + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) list[wave][pos] = j;
    carry += __popcll(mask);
    if (j < p.M && !keep && p.zero_gcol) {   // a dropped sample contributes nothing: its colour outputs read as zero
      const long pt = ray * p.M + j;
      const f4 z4 = {0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f4*>(p.zero_gcol)[pt] = z4;
      if (p.zero_relit) reinterpret_cast<f4*>(p.zero_relit)[pt] = z4;
      if (p.zero_delta) { p.zero_delta[pt * 3] = 0.f; p.zero_delta[pt * 3 + 1] = 0.f; p.zero_delta[pt * 3 + 2] = 0.f; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int off = p.offsets[ray];
  for (int k = lane; k < carry; k += 64) p.idx[off + k] = (int)(ray * p.M + list[wave][k]);
  if (p.featx_c == nullptr) return;
  const int nf4 = p.ldfx / 4;
  for (int k = 0; k < carry; ++k) {
    const long src = ray * p.M + list[wave][k];
    const long dst = off + k;
    for (int c = lane; c < nf4; c += 64)
      reinterpret_cast<f4*>(p.featx_c + dst * p.ldfx)[c] = reinterpret_cast<const f4*>(p.featx + src * p.ldfx)[c];
    if (lane < kAux / 4) reinterpret_cast<f4*>(p.aux_c + dst * kAux)[lane] = reinterpret_cast<const f4*>(p.aux + src * kAux)[lane];
  }
}
void be_prune_gather(const PruneGather& p, cnr_stream s) {
  TimingScope ts_("prune_gather", 2, 0, p.R, 0, 0, 0, s);
  hipLaunchKernelGGL(prune_gather_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("prune_gather");
}

__global__ __launch_bounds__(256) void prune_scatter_kernel(const PruneScatter p) {
  const long n = *p.count;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long pt = p.idx[i];
    reinterpret_cast<f4*>(p.gcol)[pt] = reinterpret_cast<const f4*>(p.gcol_c)[i];
    if (p.relit) reinterpret_cast<f4*>(p.relit)[pt] = reinterpret_cast<const f4*>(p.relit_c)[i];
    if (p.delta) for (int c = 0; c < 3; ++c) p.delta[pt * 3 + c] = p.delta_c[i * 3 + c];
  }
}
void be_prune_scatter(const PruneScatter& p, cnr_stream s) {
  long blocks = (p.P + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  TimingScope ts_("prune_scatter", 2, 0, p.P, 0, 0, 0, s);
  hipLaunchKernelGGL(prune_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("prune_scatter");
}

// d rays: one wavefront per ray
__global__ __launch_bounds__(256) void rays_grad_finish_kernel(const RaysGradFinish p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * 4 + wave;
  if (ray >= p.R) return;
  const int M = p.M;
  float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
  const int npe = p.multires_view > 0 ? 3 + 6 * p.multires_view : 3;
  float spe[27];
  for (int q = 0; q < 27; ++q) spe[q] = 0.0f;
  float dr[3];
  for (int k = 0; k < 3; ++k) dr[k] = p.d[ray * 3 + k];
  float snear = 0.0f, sfar = 0.0f;
  for (int j = lane; j < M; j += 64) {
    const long pt = ray * M + j;
    const float z0 = p.z[pt];
    const float dist = j + 1 < M ? p.z[pt + 1] - z0 : p.sample_dist;
    const float mid = z0 + dist * 0.5f;
    float dmid = 0.0f;
    for (int k = 0; k < 3; ++k) { float pb = p.pbar[pt * 4 + k]; so[k] += pb; sd[k] += pb * mid; dmid += pb * dr[k]; }
    for (int q = 0; q < npe; ++q) {
      float v = 0.0f;
      if (p.daux_dir_c) v += p.daux_dir_c[pt * p.lddir + 6 + q];
      if (p.daux_dir_r) v += p.daux_dir_r[pt * p.lddir + 6 + q];
      spe[q] += v;
    }
    if (p.dz_parts) {
      // mid_j = z_j + dist_j / 2, dist_j = z_{j+1} - z_j (the last section has the constant length 2/S): total cotangent of z_j
      float dz = p.dz_parts[pt * 2] + dmid;
      if (j + 1 < M) dz -= p.dz_parts[pt * 2 + 1] + 0.5f * dmid;
      if (j > 0) {
        float dmid_prev = 0.0f;
        for (int k = 0; k < 3; ++k) dmid_prev += p.pbar[(pt - 1) * 4 + k] * dr[k];
        dz += p.dz_parts[(pt - 1) * 2 + 1] + 0.5f * dmid_prev;
      }
      const float lin = linspace_at(0.0f, 1.0f, M, j);
      snear += dz * (1.0f - lin);
      sfar += dz * lin;
    }
  }
  for (int k = 0; k < 3; ++k) { so[k] = wave_sum(so[k]); sd[k] = wave_sum(sd[k]); }
  for (int q = 0; q < npe; ++q) spe[q] = wave_sum(spe[q]);
  snear = wave_sum(snear); sfar = wave_sum(sfar);
  if (lane == 0) {
    if (p.d_o && p.d_d)
      for (int k = 0; k < 3; ++k) {
        const float dk = dr[k];
        float acc = sd[k] + p.d_rays_d_alpha[ray * 3 + k] + spe[k];
        float f = 1.0f;
        for (int m = 0; m < p.multires_view; ++m) {
          acc += f * (cosf(dk * f) * spe[3 + 6 * m + k] - sinf(dk * f) * spe[6 + 6 * m + k]);
          f *= 2.0f;
        }
        p.d_d[ray * 3 + k] = acc;
        p.d_o[ray * 3 + k] = so[k];
      }
    if (p.dz_parts && p.d_near && p.d_far) { p.d_near[ray] = snear; p.d_far[ray] = sfar; }
  }
}
void be_rays_grad_finish(const RaysGradFinish& p, cnr_stream s) {
  TimingScope ts_("rays_grad_finish", 2, 0, p.R, 0, 0, 0, s);
  hipLaunchKernelGGL(rays_grad_finish_kernel, dim3((unsigned)((p.R + 3) / 4)), dim3(256), 0, s, p);
  CNR_LAUNCH_CHECK("rays_grad_finish");
}

// ------------------------------------------------------------------------------------------------
// per-parameter clip + Adam (53 tensors, 1 M floats): two launches over 4096-element chunks instead of ~150 torch launches.
//   1. per-chunk sum of squared gradient entries (fixed order inside a chunk);
//   2. every chunk's workgroup folds its OWN tensor's chunk sums in chunk order (<= 17 values) -> clip coefficient -> Adam update.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int adam_find_tensor(const AdamArgs& a, int chunk) {
  int lo = 0, hi = a.count - 1;     // last tensor whose first chunk is <= chunk (binary search: 6 dependent scalar loads, not 53)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.t[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void clip_norm_kernel(const AdamArgs a) {
  __shared__ float red[4];
  const int chunk = blockIdx.x;
  const AdamTensor t = a.t[adam_find_tensor(a, chunk)];
  const long i0 = (long)(chunk - t.chunk0) * kAdamChunk;
  float ss = 0.0f;
  for (int j = threadIdx.x; j < kAdamChunk; j += 256) { const long i = i0 + j; if (i < t.n) { const float g = t.g[i]; ss += g * g; } }
  ss = block_sum_256(ss, red);
  if (threadIdx.x == 0) a.partial[chunk] = ss;
}
__global__ __launch_bounds__(256) void clip_adam_kernel(const AdamArgs a) {
  const int chunk = blockIdx.x;
  const AdamTensor t = a.t[adam_find_tensor(a, chunk)];
  float coef = 1.0f;
  if (a.max_norm > 0.0f) {
    // the tensor's chunk sums (<= 17 for the largest layer): one load per lane, folded in a fixed order that is the same in every
    // workgroup of the tensor (a serial loop of dependent loads costs one L2 round trip per chunk)
    __shared__ float coef_s;
    const int nch = (int)((t.n + kAdamChunk - 1) / kAdamChunk);
    if (threadIdx.x < 64) {
      float part = 0.0f;
      for (int c = threadIdx.x; c < nch; c += 64) part += a.partial[t.chunk0 + c];
      part = wave_sum(part);
      if (threadIdx.x == 0) coef_s = adam_clip_coef(part, a.max_norm);
    }
    __syncthreads();
    coef = coef_s;
  }
  const long i0 = (long)(chunk - t.chunk0) * kAdamChunk;
  for (int j = threadIdx.x; j < kAdamChunk; j += 256) {
    const long i = i0 + j;
    if (i < t.n) adam_update1(a, t.w + i, t.g[i] * coef, a.m + t.off + i, a.v + t.off + i);
  }
}
void be_clip_adam(const AdamArgs& a, cnr_stream s) {
  if (a.count <= 0 || a.nchunks <= 0) return;
  TimingScope ts_("clip_adam", 2, 0, a.count, 0, 0, 0, s);
  if (a.max_norm > 0.0f) hipLaunchKernelGGL(clip_norm_kernel, dim3(a.nchunks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(a.nchunks), dim3(256), 0, s, a);
  CNR_LAUNCH_CHECK("clip_adam");
}

void be_gen_rays(const GenRays& p, cnr_stream s) { gen_rays_launch(p, p.n, s); }

// backward of the ray generator: one workgroup per camera folds the contributions of that camera's rays (thread-strided partial
// sums, then a fixed-order tree: bitwise deterministic, no float atomics); the focal gradient is summed over the cameras in order.
__global__ __launch_bounds__(256) void gen_rays_bwd_kernel(const GenRaysBwd q) {
  __shared__ float red[4];
  const int cam = blockIdx.x, tid = threadIdx.x;
  float acc[14];
  for (int k = 0; k < 14; ++k) acc[k] = 0.0f;
  for (long i = tid; i < q.f.n; i += 256) {
    int c; float out[14];
    body_gen_rays_bwd1(q, i, &c, out);
    if (c == cam) for (int k = 0; k < 14; ++k) acc[k] += out[k];
  }
  for (int k = 0; k < 14; ++k) {
    const float v = block_sum_256(acc[k], red);
    if (tid == 0) { if (k < 12) q.d_c2w[cam * 16 + k] = v; else q.d_focal_partial[cam * 2 + (k - 12)] = v; }
    __syncthreads();
  }
  if (tid < 4) q.d_c2w[cam * 16 + 12 + tid] = 0.0f;
}
__global__ void gen_rays_focal_fold_kernel(const GenRaysBwd q) {
  if (threadIdx.x < 2) {
    float s = 0.0f;
    for (int c = 0; c < q.f.n_cams; ++c) s += q.d_focal_partial[c * 2 + threadIdx.x];
    q.d_focal[threadIdx.x] = s;
  }
}
void be_gen_rays_bwd(const GenRaysBwd& q, cnr_stream s) {
  TimingScope ts_("gen_rays_bwd", 2, 0, q.f.n, 0, 0, 0, s);
  hipLaunchKernelGGL(gen_rays_bwd_kernel, dim3(q.f.n_cams), dim3(256), 0, s, q);
  hipLaunchKernelGGL(gen_rays_focal_fold_kernel, dim3(1), dim3(64), 0, s, q);
  CNR_LAUNCH_CHECK("gen_rays_bwd");
}

// ------------------------------------------------------------------------------------------------
// marching cubes on the device lattice: the 512^3 volume never leaves HBM between extract_fields and the mesh
// ------------------------------------------------------------------------------------------------
}  // namespace cnr
#define CNR_MC_QUAL static __constant__ const
#include "cnr_mc_table.h"
namespace cnr {

__global__ __launch_bounds__(256) void mc_count_kernel(const McVolume m) {
  const long n = (long)m.res * m.res * m.res;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long)gridDim.x * 256) {
    const int z = (int)(v % m.res), y = (int)((v / m.res) % m.res), x = (int)(v / ((long)m.res * m.res));
    unsigned char fl;
    const int idx = mc_cell(m.u, m.res, m.thr, x, y, z, &fl);
    m.flags[v] = fl;
    m.counts[v * 2] = mc_popcount3(fl);
    m.counts[v * 2 + 1] = idx >= 0 ? kMcNumTris[idx] : 0;
  }
}
// exclusive scan of the two count channels: per-block sums, one workgroup scans the block sums, blocks rescan with their offset
__global__ __launch_bounds__(256) void mc_block_sum_kernel(const McVolume m, long n) {
  __shared__ int red[2][4];
  const long base = (long)blockIdx.x * kMcScanBlock;
  int a = 0, b = 0;
  for (int j = threadIdx.x; j < kMcScanBlock; j += 256) { const long v = base + j; if (v < n) { a += m.counts[v * 2]; b += m.counts[v * 2 + 1]; } }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    m.block_sums[blockIdx.x * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    m.block_sums[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ __launch_bounds__(1024) void mc_scan_sums_kernel(const McVolume m, int nblocks) {
  __shared__ int wsum[2][16];
  __shared__ int carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2) carry[tid] = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + tid;
    int x[2] = {i < nblocks ? m.block_sums[i * 2] : 0, i < nblocks ? m.block_sums[i * 2 + 1] : 0};
    int incl[2] = {x[0], x[1]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y0 = __shfl_up(incl[0], d), y1 = __shfl_up(incl[1], d);
      if (lane >= d) { incl[0] += y0; incl[1] += y1; }
    }
    if (lane == 63) { wsum[0][wave] = incl[0]; wsum[1][wave] = incl[1]; }
    __syncthreads();
    int woff[2] = {0, 0};
    for (int w = 0; w < wave; ++w) { woff[0] += wsum[0][w]; woff[1] += wsum[1][w]; }
    const int c0 = carry[0], c1 = carry[1];
    if (i < nblocks) { m.block_sums[i * 2] = c0 + woff[0] + incl[0] - x[0]; m.block_sums[i * 2 + 1] = c1 + woff[1] + incl[1] - x[1]; }
    __syncthreads();
    if (tid == 1023) { carry[0] = c0 + woff[0] + incl[0]; carry[1] = c1 + woff[1] + incl[1]; }
    __syncthreads();
  }
  if (tid == 0) { m.totals[0] = carry[0]; m.totals[1] = carry[1]; }
}
__global__ __launch_bounds__(256) void mc_scan_apply_kernel(const McVolume m, long n) {
  // one workgroup rescans its 2048-element chunk: 8 elements per thread, wave scan, wave offsets through LDS
  __shared__ int wsum[2][4];
  const long base = (long)blockIdx.x * kMcScanBlock + (long)threadIdx.x * 8;
  int a[8], b[8], sa = 0, sb = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const long v = base + j; a[j] = v < n ? m.counts[v * 2] : 0; b[j] = v < n ? m.counts[v * 2 + 1] : 0; sa += a[j]; sb += b[j]; }
  int ia = sa, ib = sb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int ya = __shfl_up(ia, d), yb = __shfl_up(ib, d); if (lane >= d) { ia += ya; ib += yb; } }
  if (lane == 63) { wsum[0][wave] = ia; wsum[1][wave] = ib; }
  __syncthreads();
  int oa = m.block_sums[blockIdx.x * 2] + ia - sa, ob = m.block_sums[blockIdx.x * 2 + 1] + ib - sb;
  for (int w = 0; w < wave; ++w) { oa += wsum[0][w]; ob += wsum[1][w]; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const long v = base + j; if (v < n) { m.counts[v * 2] = oa; m.counts[v * 2 + 1] = ob; } oa += a[j]; ob += b[j]; }
}
void be_mc_count(const McVolume& m, cnr_stream s) {
  const long n = (long)m.res * m.res * m.res;
  const int nblocks = (int)((n + kMcScanBlock - 1) / kMcScanBlock);
  TimingScope ts_("mc_count_scan", 2, 0, n, 0, 0, 0, s, (double)n * (4.0 * 4.0 + 1.0 + 8.0 * 3.0));
  long cb = (n + 255) / 256; if (cb > 65535 * 4) cb = 65535 * 4;
  hipLaunchKernelGGL(mc_count_kernel, dim3((unsigned)cb), dim3(256), 0, s, m);
  hipLaunchKernelGGL(mc_block_sum_kernel, dim3(nblocks), dim3(256), 0, s, m, n);
  hipLaunchKernelGGL(mc_scan_sums_kernel, dim3(1), dim3(1024), 0, s, m, nblocks);
  hipLaunchKernelGGL(mc_scan_apply_kernel, dim3(nblocks), dim3(256), 0, s, m, n);
  CNR_LAUNCH_CHECK("mc_count");
}

struct McEmit { McVolume m; float bmin[3], bmax[3]; float* verts; int* tris; };
__global__ __launch_bounds__(256) void mc_emit_kernel(const McEmit e) {
  const McVolume& m = e.m;
  const int res = m.res;
  const long r2 = (long)res * res, n = r2 * res;
  const long stride[3] = {r2, (long)res, 1};
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long)gridDim.x * 256) {
    const unsigned fl = m.flags[v];
    const int xyz[3] = {(int)(v / r2), (int)((v / res) % res), (int)(v % res)};
    if (fl) {   // the vertices on the (up to three) level-crossing edges this voxel owns
      int k = m.counts[v * 2];
      const float u0 = m.u[v];
      for (int a = 0; a < 3; ++a) {
        if (!((fl >> a) & 1)) continue;
        const float u1 = m.u[v + stride[a]];
        const float t = (m.thr - u0) / (u1 - u0);
        for (int c = 0; c < 3; ++c) {
          const float g = (float)xyz[c] + (c == a ? t : 0.0f);      // lattice index coordinates, like mcubes' vertices
          e.verts[(long)k * 3 + c] = g / ((float)res - 1.0f) * (e.bmax[c] - e.bmin[c]) + e.bmin[c];   // NeuS.py:36-39
        }
        ++k;
      }
    }
    if (xyz[0] + 1 < res && xyz[1] + 1 < res && xyz[2] + 1 < res) {
      int idx = 0;
      for (int c = 0; c < 8; ++c) idx |= (m.u[v + (c & 1) * r2 + ((c >> 1) & 1) * res + ((c >> 2) & 1)] > m.thr ? 1 : 0) << c;
      const int nt = kMcNumTris[idx];
      int t0 = m.counts[v * 2 + 1];
      for (int t = 0; t < nt; ++t) {
        for (int q = 0; q < 3; ++q) {
          const int ed = kMcTris[idx][t * 3 + q], a = ed >> 2, kk = ed & 3;
          const int o0 = a == 0 ? 1 : 0, o1 = a == 2 ? 1 : 2;       // the two other axes in increasing order
          const long vo = v + (kk & 1) * stride[o0] + (kk >> 1) * stride[o1];
          e.tris[(long)(t0 + t) * 3 + q] = m.counts[vo * 2] + mc_popcount3(m.flags[vo] & ((1u << a) - 1u));
        }
      }
    }
  }
}
void be_mc_emit(const McVolume& m, const float* bmin, const float* bmax, float* verts, int* tris, cnr_stream s) {
  McEmit e;
  e.m = m; e.verts = verts; e.tris = tris;
  for (int c = 0; c < 3; ++c) { e.bmin[c] = bmin[c]; e.bmax[c] = bmax[c]; }
  const long n = (long)m.res * m.res * m.res;
  TimingScope ts_("mc_emit", 2, 0, n, 0, 0, 0, s);
  long cb = (n + 255) / 256; if (cb > 65535 * 4) cb = 65535 * 4;
  hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)cb), dim3(256), 0, s, e);
  CNR_LAUNCH_CHECK("mc_emit");
}

void be_grid_points(float*, cnr_stream) {}

}  // namespace cnr
