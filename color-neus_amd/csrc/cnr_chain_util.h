// Device helpers shared by the chain-fused kernels (cnr_chain.hip: value-only SDF chain; cnr_chain_fwd.hip: the saving forward chains).
#pragma once
#include <hip/hip_runtime.h>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"

namespace cnr {

constexpr int CH_ALD = 256 * 2 + 16;   // bytes per LDS row of one plane (+16: conflict-free ds_read_b128 of the fragments)

// ------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------
// exact power-of-two scale that lifts a row's largest |element| into the top f16 binade (same rule as WS_PUT_SET in cnr_gemm_ws.h)
__device__ __forceinline__ float chain_row_scale(float mx) {
  float sc = 1.0f;
  if (mx > 0.0f && mx < 3.0e38f) {
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 126;   // frexpf exponent (subnormals land below the clamp)
    if (e < -100) e = -100;
    sc = __uint_as_float((unsigned)(127 + 14 - e) << 23);
  }
  return sc;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global access (s_waitcnt
// vmcnt(0)), i.e. it would wait for the weight blocks prefetched for the next layer at both barriers of every layer.  Global data
// written by one wave is never read by another wave of the same launch, so no global ordering is needed.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two elements per VALU issue).  The epilogue of a layer runs
// with every wave of the workgroup in the same phase (the barriers keep MFMA and epilogue phases in lockstep), so its VALU
// instruction count is wall time: measured on 2 M points, 6.57 ms = 3.44 ms of MFMA + 3.13 ms of everything else.
// (tools/probes/overlap_probe.hip: VALU work of one wave does hide behind the MFMAs of the OTHER wave of its SIMD -- 12 MFMAs 223 ns,
// 96 VALU 334 ns, both 386 ns -- but two workgroups per CU running the same phases at the same time gain nothing from it.)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_splat(float x) { f2 r = {x, x}; return r; }

// nn.Softplus(beta=100, threshold=20) on two elements: max(z,0) + log2(1 + 2^(-|100 z| / ln 2)) * ln 2 / 100 (see cnr_common.h)
__device__ __forceinline__ f2 softplus100_pk(f2 z) {
  const f2 m = z * pk_splat(144.26950408889634f);
  f2 e;
  e.x = __builtin_amdgcn_exp2f(-fabsf(m.x)); e.y = __builtin_amdgcn_exp2f(-fabsf(m.y));   // (the sign / abs modifiers are free)
  const f2 s = e + pk_splat(1.0f);
  f2 l, r;
  l.x = __builtin_amdgcn_logf(s.x); l.y = __builtin_amdgcn_logf(s.y);
  r.x = fmaxf(z.x, 0.0f); r.y = fmaxf(z.y, 0.0f);
  return pk_fma(l, pk_splat(0.0069314718055994531f), r);
}

// 4 consecutive columns of one row -> hi / lo f16 planes (the per-layer kernel's ws_put4)
__device__ __forceinline__ void chain_put4(const f4& v, float sc, unsigned char* dst, int aplane) {
  f16x4 h1, h2;
  float x;
  x = v.x * sc; h1[0] = (_Float16)x; h2[0] = (_Float16)(x - (float)h1[0]);
  x = v.y * sc; h1[1] = (_Float16)x; h2[1] = (_Float16)(x - (float)h1[1]);
  x = v.z * sc; h1[2] = (_Float16)x; h2[2] = (_Float16)(x - (float)h1[2]);
  x = v.w * sc; h1[3] = (_Float16)x; h2[3] = (_Float16)(x - (float)h1[3]);
  *reinterpret_cast<f16x4*>(dst) = h1;
  *reinterpret_cast<f16x4*>(dst + aplane) = h2;
}

// 16 consecutive columns of one row -> hi / lo f16 planes (two 16-byte LDS stores per plane)
__device__ __forceinline__ void chain_put16(const f32x16& a, float sc, unsigned char* dst, int aplane) {
  h2 hi[8], lo[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f2 x = {a[2 * q], a[2 * q + 1]};
    x = x * pk_splat(sc);
    hi[q] = __builtin_convertvector(x, h2);
    const f2 back = __builtin_convertvector(hi[q], f2);
    lo[q] = __builtin_convertvector(x - back, h2);
  }
  f16x8 h1a = {hi[0][0], hi[0][1], hi[1][0], hi[1][1], hi[2][0], hi[2][1], hi[3][0], hi[3][1]};
  f16x8 h1b = {hi[4][0], hi[4][1], hi[5][0], hi[5][1], hi[6][0], hi[6][1], hi[7][0], hi[7][1]};
  f16x8 h2a = {lo[0][0], lo[0][1], lo[1][0], lo[1][1], lo[2][0], lo[2][1], lo[3][0], lo[3][1]};
  f16x8 h2b = {lo[4][0], lo[4][1], lo[5][0], lo[5][1], lo[6][0], lo[6][1], lo[7][0], lo[7][1]};
  *reinterpret_cast<f16x8*>(dst) = h1a;
  *reinterpret_cast<f16x8*>(dst + 16) = h1b;
  *reinterpret_cast<f16x8*>(dst + aplane) = h2a;
  *reinterpret_cast<f16x8*>(dst + aplane + 16) = h2b;
}


// ------------------------------------------------------------------------------------------------
// One MFMA phase of a chain-fused kernel: NKB k16 blocks of the transposed product acc[j][rt] += W_frag(kb) x Act_frag(kb, rt) as straight-line
// code (3 x v_mfma_f32_32x32x16_f16 per product: hi x lo, lo x hi, hi x hi).
//   * the weight fragments come from a 4-deep register ring: blocks kb0 .. kb0 + 3 must already be in flight (chain_wprime); block kb + 4 is
//     requested as soon as the MFMAs of block kb have read their slot;
//   * the activation fragments of block kb + 1 are read from LDS into a second register set BEFORE the MFMAs of block kb;
//   * a scheduling fence closes every block.  Without it the compiler sinks each weight load down to its first use (load, s_waitcnt vmcnt(0),
//     MFMA -- checked in the ISA of both chain kernels), i.e. every k16 block waits for an L2 round trip and the "ring" is one block deep.
// Wf layout: [column block cb][k16 block kb (nkb_w per column block)][plane][lane][8]; wbase = this wave's first column block + lane * 8.
// ------------------------------------------------------------------------------------------------
template <int CB>
__device__ __forceinline__ void chain_wload(f16x8 (&wr1)[4][CB], f16x8 (&wr2)[4][CB], int slot, const unsigned short* wlane, int nkb_w, int kb) {
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const unsigned short* b = wlane + (long)(j * nkb_w + kb) * 1024;
    wr1[slot][j] = *reinterpret_cast<const f16x8*>(b);
    wr2[slot][j] = *reinterpret_cast<const f16x8*>(b + 512);
  }
}
// the first four blocks of a phase of n >= 1 blocks (index clamped: a shorter phase loads its last block again -- no branch between the loads)
template <int CB>
__device__ __forceinline__ void chain_wprime(f16x8 (&wr1)[4][CB], f16x8 (&wr2)[4][CB], const unsigned short* wlane, int nkb_w, int kb0, int n) {
#pragma unroll
  for (int q = 0; q < 4; ++q) chain_wload<CB>(wr1, wr2, q, wlane, nkb_w, kb0 + (q < n ? q : n - 1));
}
// KB0 / NTOT: this call covers the k16 blocks [KB0, KB0 + NKB) of a phase of NTOT blocks (a phase may be cut into several calls with a
// workgroup barrier in between: the ring slots and the refill rule follow the block index within the phase; KB0 a multiple of 4).
template <int RT, int CB, int NKB, int KB0 = 0, int NTOT = NKB>
__device__ __forceinline__ void chain_mfma_blocks(f32x16 (&acc)[CB][RT], f16x8 (&wr1)[4][CB], f16x8 (&wr2)[4][CB], const unsigned char* Ab, int aplane,
                                                  const unsigned short* wlane, int nkb_w, int kb0) {
  static_assert((KB0 & 3) == 0 && KB0 + NKB <= NTOT, "chain_mfma_blocks: block range");
  f16x8 a1[2][RT], a2[2][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    a1[0][rt] = *reinterpret_cast<const f16x8*>(Ab + rt * 32 * CH_ALD + KB0 * 32);
    a2[0][rt] = *reinterpret_cast<const f16x8*>(Ab + rt * 32 * CH_ALD + aplane + KB0 * 32);
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int cur = kb & 1, gk = KB0 + kb;
    if (kb + 1 < NKB) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        a1[cur ^ 1][rt] = *reinterpret_cast<const f16x8*>(Ab + rt * 32 * CH_ALD + (gk + 1) * 32);
        a2[cur ^ 1][rt] = *reinterpret_cast<const f16x8*>(Ab + rt * 32 * CH_ALD + aplane + (gk + 1) * 32);
      }
    }
#ifdef CNR_CHAIN_MFMA16_PROBE
    // TIMING PROBE ONLY (wrong results): the same FLOP, fragment registers and accumulator registers issued as v_mfma_f32_16x16x32_f16.  It measured the chain
    // kernels 15-27 % faster (profiles/r06_probe_chain_mfma16.txt) and MISLED: the real conversion (tools/ab/exp_chain_mfma16.patch: k32 fragment layout, a lane
    // owning two points x 8 columns, parity-green on the GPU) is 5-8 % SLOWER than this 32 x 32 x 16 form (profiles/r06_ab_chain_mfma16.txt) -- the probe's wrong
    // activations change what the matrix pipe is fed (zero-heavy operands draw far less power: tools/mfma_power.py), these kernels run below the board's power
    // limit (so cheaper MFMAs do not buy time as they do in layer_dw), and the 16 x 16 lane ownership costs the epilogue two cross-lane steps per row maximum.
#define PROBE16(W, A)                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < CB; ++j) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                      \
      typedef float pf4 __attribute__((ext_vector_type(4)));                                                                 \
      f32x16& c = acc[j][rt];                                                                                                \
      if (gk & 1) {                                                                                                          \
        pf4 s0 = {c[8], c[9], c[10], c[11]}, s1 = {c[12], c[13], c[14], c[15]};                                              \
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[gk & 3][j], A[cur][rt], s0, 0, 0, 0);                                  \
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[gk & 3][j], A[cur][(rt + 1) % RT], s1, 0, 0, 0);                       \
        c[8] = s0[0]; c[9] = s0[1]; c[10] = s0[2]; c[11] = s0[3]; c[12] = s1[0]; c[13] = s1[1]; c[14] = s1[2]; c[15] = s1[3]; \
      } else {                                                                                                               \
        pf4 s0 = {c[0], c[1], c[2], c[3]}, s1 = {c[4], c[5], c[6], c[7]};                                                    \
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[gk & 3][j], A[cur][rt], s0, 0, 0, 0);                                  \
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[gk & 3][j], A[cur][(rt + 1) % RT], s1, 0, 0, 0);                       \
        c[0] = s0[0]; c[1] = s0[1]; c[2] = s0[2]; c[3] = s0[3]; c[4] = s1[0]; c[5] = s1[1]; c[6] = s1[2]; c[7] = s1[3];       \
      }                                                                                                                      \
    }
    PROBE16(wr2, a1)
    PROBE16(wr1, a2)
    PROBE16(wr1, a1)
#undef PROBE16
#else
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[j][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr2[gk & 3][j], a1[cur][rt], acc[j][rt], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[j][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr1[gk & 3][j], a2[cur][rt], acc[j][rt], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[j][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr1[gk & 3][j], a1[cur][rt], acc[j][rt], 0, 0, 0);
#endif
    if (gk + 4 < NTOT) chain_wload<CB>(wr1, wr2, gk & 3, wlane, nkb_w, kb0 + gk + 4);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace cnr
