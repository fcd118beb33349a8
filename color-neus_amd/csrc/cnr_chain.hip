// Chain-fused MLP kernels for gfx950 (MI355X / CDNA4): a workgroup carries a tile of points through ALL layers of a chain.
//
// Why: the per-layer weight-stationary kernel (cnr_gemm_ws.h) streams every activation through HBM (2 KB per point and layer) and
// sits at ~0.5 of the HBM peak with the matrix cores ~20 % busy.  Here the activations never leave the CU between layers:
//   * the tile's current activations live in LDS as two f16 planes (hi + lo of the exactly power-of-two scaled fp32 row: the same
//     error-free split as the per-layer kernel, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulation);
//   * the layer's weights are NOT resident: every wave streams the fragments of its 32 output columns from L2 / Infinity Cache
//     straight into registers (fragment-major layout written by pack_frags_kernel: one contiguous, fully coalesced 1 KB read per
//     k16 block and plane), 4 blocks ahead of their use -- 256 KB per layer and tile, shared by every CU of an XCD through its L2;
//   * the MFMA is issued TRANSPOSED (A operand = weights, B operand = activations): a lane then owns ONE point and 16 CONSECUTIVE
//     output columns (the weight rows are dealt to the fragment so that register r <-> column c0 + 16 * half + r), so the fused
//     epilogue (bias, softplus, skip concat, row max, f16 split) is plain per-lane code and the next layer's planes are written with
//     16-byte LDS stores -- no transpose through LDS, one exchange between the two 32-lane halves per row for the row max (a VALU lane swap, cnr_pair32_max);
//   * two barriers per layer (row-max exchange, planes ready).
// Arithmetic is the per-layer kernels' arithmetic (same split, same scales, same MFMA order; the epilogue uses fused multiply-adds),
// so results agree with the unfused path to fp32 round-off.
// Scope: the chains that save nothing (sampler, lattice, sdf()).  A variant that also stored the pre-activations, features and
// row scales for the backward pass was built and measured (commit eeb54d9: parity green, 3.8 ms against 2.4 ms for the nine
// per-layer launches at 524 288 points): in the transposed layout a lane owns a 64-byte piece of a row, a wave-wide store
// touches 32 rows, and such stores drain at ~8 B/clk/CU; with the activations saved the chain is no longer MFMA/VALU bound either
// way (1 KB written per point and layer against 2 KB moved by the per-layer kernel), so the saving chains stay per layer.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"
#include "cnr_chain_util.h"

namespace cnr {

// ------------------------------------------------------------------------------------------------
// fragment-major weight planes: Wf[cb][kb][plane][lane][8] = plane[n(cb, lane & 31)][kb * 16 + (lane >> 5) * 8 + 0..7]
// with n(cb, m) = cb * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3): MFMA row m of the transposed product -> layer column
// ------------------------------------------------------------------------------------------------
constexpr int kPackBatch = 64;
struct PackBatch { int count; int blk_start[kPackBatch + 1]; PackJob j[kPackBatch]; };
static_assert(sizeof(PackBatch) <= 4096, "kernel argument block");

__global__ __launch_bounds__(64) void pack_frags_kernel(const PackBatch b) {
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.blk_start[i + 1]) ++i;
  const PackJob& j = b.j[i];
  const int blk = (int)blockIdx.x - b.blk_start[i];        // (cb * nkb + kb) * 2 + plane
  const int nkb = j.ld / 16;
  const int plane = blk & 1, kb = (blk >> 1) % nkb, cb = (blk >> 1) / nkb;
  const int lane = threadIdx.x, m = lane & 31;
  const int n = cb * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < j.rows) v = *reinterpret_cast<const u16x8*>(j.planes + (long)plane * j.plane_stride + (long)n * j.ld + kb * 16 + (lane >> 5) * 8);
  *reinterpret_cast<u16x8*>(j.Wf + ((long)blk * 64 + lane) * 8) = v;
}

void be_pack_frags_many(const PackJob* jobs, int count, cnr_stream s) {
  for (int i0 = 0; i0 < count; i0 += kPackBatch) {
    PackBatch b;
    b.count = count - i0 < kPackBatch ? count - i0 : kPackBatch;
    int blks = 0;
    for (int i = 0; i < b.count; ++i) { b.blk_start[i] = blks; b.j[i] = jobs[i0 + i]; blks += 8 * (jobs[i0 + i].ld / 16) * 2; }
    b.blk_start[b.count] = blks;
    TimingScope ts_("pack_frags", 2, 0, blks, 0, 0, 0, s);
    hipLaunchKernelGGL(pack_frags_kernel, dim3(blks), dim3(64), 0, s, b);
  }
  CNR_LAUNCH_CHECK("pack_frags");
}

// ------------------------------------------------------------------------------------------------
// SDF value chain (no-grad uses: hierarchical sampler NeuS.py:345,191; extract_fields NeuS.py:14-28; sdf()): E -> sdf, nothing saved
//
// Geometry <RT, CB>: a workgroup of 8 / CB waves owns a tile of T = 32 * RT points; a wave owns CB blocks of 32 output columns
// (12 = 3 * RT * CB MFMAs per k16 block for both shipped shapes):
//   <4, 1>  512 threads, 128 points, one workgroup per CU: least weight traffic (2 KB per point and layer from L2), but MFMA and
//           epilogue phases of the whole CU alternate in lockstep;
//   <2, 2>  256 threads, 64 points, two workgroups per CU (74 KB of LDS each) at 4 KB of L2 weight reads per point and layer.
//           Measured equal to <4, 1> (6.5 ms for 2 M points): the two workgroups start together, do identical work and stay in phase
//           (both in the MFMA phase, then both in the epilogue), also when half of them are started a few microseconds late;
//   <1, 2>  32-point tiles for small point counts (fills the chip from 8192 points).
// Tried and dropped (round 4, commit 52bf6b3): two 4-wave groups per workgroup, each on its own 64-point half tile (<2, 2> per group), group 1
// running two barriers behind group 0 so that on every SIMD one wave is in an MFMA phase while the other is in an epilogue phase -- enforced by
// the barriers (4 per layer), with the weight ring pinned four blocks deep.  Bit-identical, and no faster: 1.57 ms against 1.55 ms per step for
// the sampler's chains, the same for a delay of 0, 1, 2 or 3 barriers.  MFMA time and epilogue time add up whatever the phase relation; the
// chip runs these kernels at 1.9-2.1 GHz of its 2.4 GHz (GRBM_GUI_ACTIVE / duration), i.e. it is at its power limit either way.
// ------------------------------------------------------------------------------------------------
template <int RT, int CB>
__global__ __launch_bounds__(512 / CB, 2 / (3 - CB) + 0) void sdf_value_chain_kernel(const SdfValueChain c) {
  constexpr int WAVES = 8 / CB, THREADS = 64 * WAVES;
  constexpr int T = 32 * RT;
  constexpr int APLANE = T * CH_ALD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rs = reinterpret_cast<float*>(smem + 2 * APLANE);   // [T] 1 / row scale of the current layer input
  float* pm = rs + T;                                        // [T][8] per-wave partial row maxima / partial dot products
  float* cwb = pm + T * 8;                                   // [2][512] column scales | biases of the current / next layer
  float* wtop = cwb + 1024;                                  // [256] sdf row of the top layer
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const long ntiles = (c.P + T - 1) / T;
  // per-lane indices are re-derived from a laundered copy of the thread index in every layer iteration: hoisted out of the loops they pin
  // dozens of address registers for the whole kernel (29 - 33 spilled VGPRs before)
#define VC_LANE_INDICES                                                                      \
  int tid = tid0;                                                                            \
  asm volatile("" : "+v"(tid));                                                              \
  const int lane = tid & 63, half = lane >> 5, pt = lane & 31;                               \
  const int cbase = wave * CB * 32 + 16 * half;   /* this lane's 16 consecutive output columns of its j-th block: cbase + 32 j */

  f16x8 wr1[4][CB], wr2[4][CB];                              // weight fragment ring: 4 k16 blocks in flight
  auto wlane_of = [&](const FusedLayer& L, int lane_) __attribute__((always_inline)) { return L.Wf + (long)wave * CB * (L.K >> 4) * 1024 + lane_ * 8; };
  auto wprime = [&](const FusedLayer& L, int lane_) __attribute__((always_inline)) { chain_wprime<CB>(wr1, wr2, wlane_of(L, lane_), L.K >> 4, 0, L.K >> 4); };
  auto cw_fetch = [&](const FusedLayer& L, int tid_) __attribute__((always_inline)) {   // threads 0..127: one float4 of [column scales (256) | biases (256)]
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (tid_ < 64) v = *reinterpret_cast<const f4*>(L.wsc + tid_ * 4);
    else if (tid_ < 128) v = *reinterpret_cast<const f4*>(L.bias + (tid_ - 64) * 4);
    return v;
  };

  wprime(c.lay[0], tid0 & 63);
  for (int i = tid0; i < 256; i += THREADS) wtop[i] = c.wtop[i];
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- layer-0 input: E rows -> planes (16 threads per row, 4 columns each); all row groups are requested before the first is converted
    // (consumed one by one the compiler waits for each load before it issues the next: RT exposed round trips per tile)
    constexpr int NPASS = T * 16 / THREADS;
    {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    if (tid < 128) *reinterpret_cast<f4*>(cwb + tid * 4) = cw_fetch(c.lay[0], tid);
    f4 erows[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row_l = pass * (THREADS / 16) + (tid >> 4), sc4 = (tid & 15) * 4;
      long grow = tile * T + row_l; if (grow >= c.P) grow = c.P - 1;
      erows[pass] = *reinterpret_cast<const f4*>(c.E + grow * kEmb + (sc4 < kEmb ? sc4 : 0));
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row_l = pass * (THREADS / 16) + (tid >> 4), sc4 = (tid & 15) * 4;
      const f4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f4 v = sc4 < kEmb ? erows[pass] : z4;
      float mx = ws_absmax4(v);
      mx = cnr_max16(mx);
      const float sc = chain_row_scale(mx);
      if (sc4 < kEmb) {
        f16x4 h1, h2;
        float x;
        x = v.x * sc; h1[0] = (_Float16)x; h2[0] = (_Float16)(x - (float)h1[0]);
        x = v.y * sc; h1[1] = (_Float16)x; h2[1] = (_Float16)(x - (float)h1[1]);
        x = v.z * sc; h1[2] = (_Float16)x; h2[2] = (_Float16)(x - (float)h1[2]);
        x = v.w * sc; h1[3] = (_Float16)x; h2[3] = (_Float16)(x - (float)h1[3]);
        unsigned char* dst = smem + row_l * CH_ALD + sc4 * 2;
        *reinterpret_cast<f16x4*>(dst) = h1;
        *reinterpret_cast<f16x4*>(dst + APLANE) = h2;
      }
      if ((tid & 15) == 0) rs[row_l] = cnr_pow2_rcp(sc);
    }
    }
    lds_barrier();

    for (int l = 0; l < c.nl; ++l) {
      VC_LANE_INDICES
      const FusedLayer& L = c.lay[l];
      const int nkb = L.K >> 4;
      const bool last = l + 1 == c.nl;
      const unsigned short* wlane = wlane_of(L, lane);
      const f4 cw_next = cw_fetch(c.lay[last ? l : l + 1], tid);   // lands while the MFMAs run; parked in LDS behind the row-max barrier
      f32x16 acc[CB][RT];
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][rt][r] = 0.0f;
      const unsigned char* Ab = smem + pt * CH_ALD + half * 16;
      // The k16 blocks of a layer are straight-line code (block count pinned at compile time), weight ring four blocks deep, scheduling fence
      // per block (chain_mfma_blocks in cnr_chain_util.h)
      if (nkb == 16) chain_mfma_blocks<RT, CB, 16>(acc, wr1, wr2, Ab, APLANE, wlane, 16, 0);
      else chain_mfma_blocks<RT, CB, 3>(acc, wr1, wr2, Ab, APLANE, wlane, 3, 0);
      // the next layer's (or the next tile's first layer's) leading weight blocks travel while the epilogue runs
      wprime(last ? c.lay[0] : c.lay[l + 1], lane);

      // ---- epilogue: z = acc * (1 / row scale) * (1 / column scale) + bias ; a = softplus(z) ; skip concat ; row max
      const float* cw = cwb + (l & 1) * 512;
      const bool next_skip = !last && ((c.skip_mask >> (l + 1)) & 1);
      const float oscale = next_skip ? kInvSqrt2 : 1.0f;
      const bool ragged = L.N < 256;   // wave-uniform: only the layer in front of a skip connection (217 columns + 39 of e)
      float red[RT], dotj[CB][RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        red[rt] = 0.0f;
#pragma unroll
        for (int j = 0; j < CB; ++j) dotj[j][rt] = 0.0f;
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const int c0 = cbase + 32 * j;
        f4 wsc4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wsc4[q] = *reinterpret_cast<const f4*>(cw + c0 + 4 * q);
          b4[q] = *reinterpret_cast<const f4*>(cw + 256 + c0 + 4 * q);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          const f2 rsc = pk_splat(rs[row_l]);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f2 w = {wsc4[q >> 1][(2 * q) & 3], wsc4[q >> 1][(2 * q + 1) & 3]};
            const f2 b = {b4[q >> 1][(2 * q) & 3], b4[q >> 1][(2 * q + 1) & 3]};
            f2 a = {acc[j][rt][2 * q], acc[j][rt][2 * q + 1]};
            a = pk_fma(a, rsc * w, b);                       // z = acc / (row scale * column scale) + bias
            a = softplus100_pk(a);
            if (next_skip) a = a * pk_splat(kInvSqrt2);
            acc[j][rt][2 * q] = a.x; acc[j][rt][2 * q + 1] = a.y;
          }
          if (ragged && c0 + 16 > L.N) {
            // columns >= N (only the lanes that own them enter): [softplus(z) | e] / sqrt(2) for a skip layer (fields.py:86-87), zero
            // otherwise.  Branch-free per element: the e value is fetched from a clamped index and selected.
            long grow = tile * T + row_l; if (grow >= c.P) grow = c.P - 1;
            const float* erow = c.E + grow * kEmb;
            float ev[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int ei = c0 + r - L.N; ev[r] = erow[ei < 0 ? 0 : (ei < kEmb ? ei : kEmb - 1)]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ei = c0 + r - L.N;
              const float tail = (next_skip && ei < c.emb) ? ev[r] * oscale : 0.0f;
              acc[j][rt][r] = ei < 0 ? acc[j][rt][r] : tail;
            }
          }
          float mx = 0.0f, dot = 0.0f;
          if (last) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f4 wt = *reinterpret_cast<const f4*>(wtop + c0 + 4 * q);
              dot = fmaf(acc[j][rt][4 * q], wt.x, dot); dot = fmaf(acc[j][rt][4 * q + 1], wt.y, dot);
              dot = fmaf(acc[j][rt][4 * q + 2], wt.z, dot); dot = fmaf(acc[j][rt][4 * q + 3], wt.w, dot);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(fabsf(acc[j][rt][r]), fabsf(acc[j][rt][r + 1])), mx);
          }
          if (last) dotj[j][rt] = dot; else red[rt] = fmaxf(red[rt], mx);
        }
      }
      // partial results per row: the row maximum per wave; the sdf dot product per 32-COLUMN BLOCK (its two 16-column halves added first), slot =
      // block index -- the same 8 partial sums in the same order for every tile shape, so that the sdf of a point does not depend on how many points
      // the call holds (round 6: a 1024-ray chunk of a view used to differ from the same rays inside a 65536-ray chunk by one ulp of sdf, because
      // <2, 2> added the halves of two blocks first; bench.py now asserts that the two chunkings render the same image)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (last) {
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            const float both = cnr_pair32_sum(dotj[j][rt]);
            if (half == 0) pm[(rt * 32 + pt) * 8 + wave * CB + j] = both;
          }
        } else {
          const float both = cnr_pair32_max(red[rt]);
          if (half == 0) pm[(rt * 32 + pt) * 8 + wave] = both;
        }
      }
      lds_barrier();   // partial maxima visible; every wave is done reading the planes of this layer's input
      if (!last) {
        if (tid < 128) *reinterpret_cast<f4*>(cwb + ((l + 1) & 1) * 512 + tid * 4) = cw_next;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          float mx = pm[row_l * 8];
#pragma unroll
          for (int w = 1; w < WAVES; ++w) mx = fmaxf(mx, pm[row_l * 8 + w]);
          const float sc = chain_row_scale(mx);
#pragma unroll
          for (int j = 0; j < CB; ++j) chain_put16(acc[j][rt], sc, smem + row_l * CH_ALD + (cbase + 32 * j) * 2, APLANE);
          if (wave == 0 && half == 0) rs[row_l] = cnr_pow2_rcp(sc);
        }
      } else {
        // sdf = (softplus(z_top-1) . w_sdf + b_sdf) * top_scale: the per-wave partial sums in a fixed order
        for (int row = tid; row < T; row += THREADS) {
          float sum = pm[row * 8];
#pragma unroll
          for (int w = 1; w < 8; ++w) sum += pm[row * 8 + w];   // (8 column blocks, whatever the number of waves)
          const long grow = tile * T + row;
          if (grow < c.P) c.sdf_out[grow] = (sum + c.btop[0]) * c.top_scale;
        }
      }
      lds_barrier();
    }
  }
}

#undef VC_LANE_INDICES

template <int RT, int CB>
static void launch_sdf_value_chain(const SdfValueChain& c, cnr_stream s) {
  constexpr int T = 32 * RT, THREADS = 512 / CB;
  const size_t lds = (size_t)2 * T * CH_ALD + (size_t)T * 9 * sizeof(float) + (size_t)(1024 + 256) * sizeof(float);
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_value_chain_kernel<RT, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long ntiles = (c.P + T - 1) / T;
  const int wgs_env = debug_flags().chain_wgs;
  const long wgs = wgs_env > 0 ? wgs_env : (lds * 2 <= 160 * 1024 ? 512 : 256);   // persistent: as many workgroups as the chip holds at once
  const unsigned grid = (unsigned)(ntiles < wgs ? ntiles : wgs);
  double macs = 0.0;
  for (int l = 0; l < c.nl; ++l) macs += (double)c.lay[l].K * 256.0;
  TimingScope ts_("chain_sdf_value", 3, RT * 10 + CB, c.P, (int)(macs / 256.0), 256, 1, s, (double)c.P * (kEmb + 1) * 4.0);
  hipLaunchKernelGGL((sdf_value_chain_kernel<RT, CB>), dim3(grid), dim3(THREADS), lds, s, c);
}

bool be_sdf_value_chain(const SdfValueChain& c, cnr_stream s) {
  const bool off = debug_flags().no_fused;   // debugging aid: per-layer kernels everywhere
  if (off || c.P <= 0) return false;
  for (int l = 0; l < c.nl; ++l)
    if ((c.lay[l].K != 256 && c.lay[l].K != 48) || c.lay[l].N > 256 || c.lay[l].N < 1) return false;   // k16 block counts the kernel pins
  const int force = debug_flags().chain_shape;   // tuning aid: 41, 22, 12
  const int shape = force ? force : (c.P >= 256L * 128 ? 41 : (c.P >= 256L * 64 ? 22 : 12));
  if (shape == 41) launch_sdf_value_chain<4, 1>(c, s);
  else if (shape == 22) launch_sdf_value_chain<2, 2>(c, s);
  else launch_sdf_value_chain<1, 2>(c, s);
  CNR_LAUNCH_CHECK("chain_sdf_value");
  return true;
}

}  // namespace cnr
