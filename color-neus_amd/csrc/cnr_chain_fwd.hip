// Chain-fused SAVING forward kernels for gfx950 (MI355X / CDNA4): the ReLU stacks of the render path (colour network, relight network)
// in ONE launch.  A workgroup carries a tile of 32 * RT points through every layer of both stacks; between layers the activations stay on
// the CU as two f16 planes in LDS (the chain-fused value kernel's scheme, cnr_chain.hip: error-free hi / lo split of the exactly
// power-of-two scaled fp32 row, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulation, weights streamed from L2 in MFMA-fragment
// order four k16 blocks ahead, TRANSPOSED product so that a lane owns one point x 16 consecutive output columns).
//
// What the forward pass must leave behind for the backward pass -- every hidden layer's ReLU output (1 KB per point and layer) and the
// row scales of the layer inputs -- is what sank the first saving variant of the chain kernel (commit eeb54d9): in the transposed layout a
// lane holds a 64-byte piece of a row, a wave-wide 16-byte store touches 64 different 64-byte segments, and such stores drain at about
// 8 B/clk/CU.  Here every wave passes its 32 x 32 output block through a private 2 KB LDS buffer, 16 rows at a time (XOR-swizzled 16-byte
// chunks: conflict-free both ways), and stores it as 8 rows x 128 contiguous bytes per instruction -- full cache lines, the store pattern
// of the per-layer kernel's epilogue.
//
// Against the per-layer launches this removes, per point: the read-back of every hidden activation (7 x 1 KB), the second read of the
// 3-wide heads' input (2 x 1 KB), 9 launches, and for the narrow inputs (relight in_layer: 48 columns; colour layer 0: 256 + 16..48
// columns; relight y-layer: 256 + 3) the special launches that re-read 1 KB/point tensors for a 48-column product.
//
// Layers whose input is wider than 256 columns (colour layer 0: [feat | p g PE(v)], relight y-layer: [h | rgb]) run as TWO input
// segments through the same LDS planes: the 256 main columns, then -- one barrier later -- the 16..48 extra columns staged over them, the
// accumulators carried across.  Both segments share ONE row scale (the power of two of the whole input row, as the per-layer kernel
// forms it), so no accumulator rescaling is needed and the k16 blocks are summed in the per-layer kernel's order.
// The 3-wide heads (rgb, relight offset) are fp32 dot products of the last hidden layer's ReLU output while it is still in registers;
// rgb stays in LDS for the relight y-layer and the relight head's inverse-sigmoid combination.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"
#include "cnr_chain_util.h"

namespace cnr {

// What the forward chains store for the backward pass is read again tens of launches later: as NON-TEMPORAL stores these rows do not displace the weight fragments
// every tile streams from L2 (CH_NT_SAVE 0: plain stores, A/B builds)
#ifndef CH_NT_SAVE
#define CH_NT_SAVE 1
#endif
#if CH_NT_SAVE
#define CH_SAVE_STORE(p_, v_) __builtin_nontemporal_store((v_), reinterpret_cast<f4*>(p_))
#else
#define CH_SAVE_STORE(p_, v_) (*reinterpret_cast<f4*>(p_) = (v_))
#endif


// the scale a row's consumers get (LayerGemm::rs_out convention): 0 for an all-zero row, NaN for a non-finite one
__device__ __forceinline__ float chain_rs_value(float mx, float sc) { return (mx > 0.0f && mx < 3.0e38f) ? sc : (mx == 0.0f ? 0.0f : __builtin_nanf("")); }

template <int RT>
__global__ __launch_bounds__(512, 1) void relu_chain_fwd_kernel(const ReluChainFwd c) {
  constexpr int T = 32 * RT;
  constexpr int APLANE = T * CH_ALD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rs = reinterpret_cast<float*>(smem + 2 * APLANE);   // [T] 1 / row scale of the current step's input
  float* rsf = rs + T;                                       // [T] the row scale itself (the extra segment is staged with it)
  float* pm = rsf + T;                                       // [T][8] per-wave partial row maxima
  float* cwb = pm + T * 8;                                   // [2][512] column scales | biases of the current / next step
  float* rgbs = cwb + 1024;                                  // [T][4] output of the colour head (rgb, 0)
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // (scalar: the per-wave bases below stay in SGPRs)
  unsigned char* tb = reinterpret_cast<unsigned char*>(rgbs + T * 4) + wave * 2048;   // this wave's [16 rows][128 B] transposition buffer
  float* hp = reinterpret_cast<float*>(smem);                // [T][8][4] partial head dot products (in the planes: dead at the end of a chain)
  const long P = c.P_dev != nullptr ? (long)*c.P_dev : c.P;   // (compacted runs: only the device knows how many rows there are)
  const int* const ridx = c.row_idx;
  const long ntiles = (P + T - 1) / T;

  f16x8 wr1[4][1], wr2[4][1];                                // weight fragment ring: 4 k16 blocks in flight
  auto wlane_of = [&](const ChainFwdStep& S, int lane_) __attribute__((always_inline)) { return S.Wf + (long)wave * S.nkb_w * 1024 + lane_ * 8; };
  auto cw_fetch = [&](const ChainFwdStep& S, int tid) __attribute__((always_inline)) {      // threads 0..127: one float4 of [column scales (256) | biases (256)]
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (tid < 64) v = *reinterpret_cast<const f4*>(S.wsc + tid * 4);
    else if (tid < 128) v = *reinterpret_cast<const f4*>(S.bias + (tid - 64) * 4);
    return v;
  };

  chain_wprime<1>(wr1, wr2, wlane_of(c.st[0], tid0 & 63), c.st[0].nkb_w, 0, c.st[0].nkb_main);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (tid0 < 128) *reinterpret_cast<f4*>(cwb + tid0 * 4) = cw_fetch(c.st[0], tid0);
    for (int s = 0; s < c.nsteps; ++s) {
      // The thread index is laundered once per step: every per-lane LDS / global offset below is derived from it, so the compiler recomputes
      // them per step (a few integer operations) instead of hoisting dozens of loop-invariant addresses out of the loops and spilling them.
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, half = lane >> 5, pt = lane & 31;
      const int cbase = wave * 32 + 16 * half;               // this lane's 16 consecutive output columns
      const ChainFwdStep& S = c.st[s];
      const bool last_step = s + 1 == c.nsteps;
      const ChainFwdStep& Sn = c.st[last_step ? 0 : s + 1];
      const unsigned short* wlane = wlane_of(S, lane);
      const f4 cw_next = cw_fetch(Sn, tid);                       // lands while the MFMAs run; parked in LDS behind the row-max barrier
      float hbias[3] = {0.f, 0.f, 0.f};                           // a head follows this step: its biases now (same reason as for the head's weight rows below)
      if (S.head != 0) {
        const ChainFwdHead& Hh = S.head == 1 ? c.col_head : c.rel_head;
#pragma unroll
        for (int j = 0; j < 3; ++j) hbias[j] = Hh.bias[j < Hh.n ? j : 0];
      }
      // ---- a chain starts: its input rows HBM -> registers -> exact row scale -> f16 planes (16 threads per row, 4 columns each)
      f4 vx[RT];   // the extra input columns of the chain start's rows (x_src == 1): loaded with the row, staged after the main MFMA phase
#pragma unroll
      for (int pass = 0; pass < RT; ++pass) vx[pass] = f4{0.f, 0.f, 0.f, 0.f};
      if (S.in != nullptr) {
        const int ncol = 16 * S.nkb_main, xcol = S.x_src == 1 ? 16 * S.nkb_x : 0;
        const long tile0 = tile * T;
        const int rows_left = (int)((P - tile0) < T ? (P - tile0) : T);              // rows of this tile that exist (>= 1)
        const float* in_tile = S.in + (ridx != nullptr ? 0 : tile0 * S.ld_in);       // (uniform: scalar base + 32-bit lane offsets)
        // all loads of the RT row groups first (one exposed memory round trip per chain start, not RT), then the conversion
        const int sc4 = (tid & 15) * 4;
        const f4 z4 = {0.f, 0.f, 0.f, 0.f};
        f4 v0[RT], v1[RT], v2[RT], v3[RT];
#pragma unroll
        for (int pass = 0; pass < RT; ++pass) {
          const int row_l = pass * 32 + (tid >> 4);
          const int row_c = row_l < rows_left ? row_l : rows_left - 1;               // rows past the end read the last row (never stored)
          const float* src = ridx != nullptr ? in_tile + (long)ridx[tile0 + row_c] * S.ld_in : in_tile + row_c * S.ld_in;
          v0[pass] = z4; v1[pass] = z4; v2[pass] = z4; v3[pass] = z4; vx[pass] = z4;
          if (sc4 < ncol) v0[pass] = *reinterpret_cast<const f4*>(src + sc4);
          if (64 + sc4 < ncol) v1[pass] = *reinterpret_cast<const f4*>(src + 64 + sc4);
          if (128 + sc4 < ncol) v2[pass] = *reinterpret_cast<const f4*>(src + 128 + sc4);
          if (192 + sc4 < ncol) v3[pass] = *reinterpret_cast<const f4*>(src + 192 + sc4);
          if (sc4 < xcol) vx[pass] = *reinterpret_cast<const f4*>(src + ncol + sc4);   // (here only its magnitude is used: one scale for the whole row)
        }
#pragma unroll
        for (int pass = 0; pass < RT; ++pass) {
          const int row_l = pass * 32 + (tid >> 4);
          float mx = fmaxf(fmaxf(fmaxf(ws_absmax4(v0[pass]), ws_absmax4(v1[pass])), fmaxf(ws_absmax4(v2[pass]), ws_absmax4(v3[pass]))), ws_absmax4(vx[pass]));
          mx = cnr_max16(mx);
          const float sc = chain_row_scale(mx);
          unsigned char* dst = smem + row_l * CH_ALD + sc4 * 2;
          if (sc4 < ncol) chain_put4(v0[pass], sc, dst, APLANE);
          if (64 + sc4 < ncol) chain_put4(v1[pass], sc, dst + 128, APLANE);
          if (128 + sc4 < ncol) chain_put4(v2[pass], sc, dst + 256, APLANE);
          if (192 + sc4 < ncol) chain_put4(v3[pass], sc, dst + 384, APLANE);
          if ((tid & 15) == 0) {
            rs[row_l] = cnr_pow2_rcp(sc); rsf[row_l] = sc;
            if (S.rs_in != nullptr && row_l < rows_left) S.rs_in[tile0 + row_l] = chain_rs_value(mx, sc);
          }
        }
        lds_barrier();
      }

      f32x16 acc1[1][RT];
      f32x16 (&acc)[RT] = acc1[0];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
      const unsigned char* Ab = smem + pt * CH_ALD + half * 16;
      // k16 blocks as straight-line code, ring four blocks deep, one scheduling fence per block (chain_mfma_blocks); kb0: first block of the phase in W
      auto mfma_phase = [&](int nkb, int kb0) __attribute__((always_inline)) {
        if (nkb == 16) chain_mfma_blocks<RT, 1, 16>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, kb0);
        else if (nkb == 3) chain_mfma_blocks<RT, 1, 3>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, kb0);
        else if (nkb == 2) chain_mfma_blocks<RT, 1, 2>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, kb0);
        else chain_mfma_blocks<RT, 1, 1>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, kb0);
      };
      if (!(CNR_ABLATION(c.dbg) & 2)) mfma_phase(S.nkb_main, 0);
      if (S.nkb_x > 0) {
        // ---- the extra input columns, staged over the main ones with the SAME row scale
        chain_wprime<1>(wr1, wr2, wlane, S.nkb_w, S.nkb_main, S.nkb_x);
        lds_barrier();                                        // every wave is done reading the main segment
#pragma unroll
        for (int pass = 0; pass < RT; ++pass) {
          const int row_l = pass * 32 + (tid >> 4), sc4 = (tid & 15) * 4;
          if (sc4 < 16 * S.nkb_x) {
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (S.x_src == 1) {
              v = vx[pass];                                   // (same thread, same row, same columns as at the chain start)
            } else if (sc4 == 0) {
              v = *reinterpret_cast<const f4*>(rgbs + row_l * 4);
            }
            chain_put4(v, rsf[row_l], smem + row_l * CH_ALD + sc4 * 2, APLANE);
          }
        }
        lds_barrier();
        if (!(CNR_ABLATION(c.dbg) & 2)) {
          if (S.nkb_x == 3) chain_mfma_blocks<RT, 1, 3>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, S.nkb_main);
          else if (S.nkb_x == 2) chain_mfma_blocks<RT, 1, 2>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, S.nkb_main);
          else chain_mfma_blocks<RT, 1, 1>(acc1, wr1, wr2, Ab, APLANE, wlane, S.nkb_w, S.nkb_main);
        }
      }
      // the next step's (or the next tile's first step's) leading weight blocks travel while the epilogue runs
      chain_wprime<1>(wr1, wr2, wlane_of(Sn, lane), Sn.nkb_w, 0, Sn.nkb_main);

      // ---- epilogue: a = relu(acc / (row scale * column scale) + bias) ; row max
      const float* cw = cwb + (s & 1) * 512;
      const bool chain_end = S.head != 0;                     // a head follows: the next step (if any) starts from global rows
      {
        f4 wsc4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wsc4[q] = *reinterpret_cast<const f4*>(cw + cbase + 4 * q);
          b4[q] = *reinterpret_cast<const f4*>(cw + 256 + cbase + 4 * q);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          const f2 rsc = pk_splat(rs[row_l]);
          float mx = 0.0f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f2 w = {wsc4[q >> 1][(2 * q) & 3], wsc4[q >> 1][(2 * q + 1) & 3]};
            const f2 b = {b4[q >> 1][(2 * q) & 3], b4[q >> 1][(2 * q + 1) & 3]};
            f2 a = {acc[rt][2 * q], acc[rt][2 * q + 1]};
            a = pk_fma(a, rsc * w, b);
            a.x = fmaxf(a.x, 0.0f); a.y = fmaxf(a.y, 0.0f);
            acc[rt][2 * q] = a.x; acc[rt][2 * q + 1] = a.y;
            mx = fmaxf(fmaxf(a.x, a.y), mx);
          }
          if (!chain_end) {
            const float both = cnr_pair32_max(mx);
            if (half == 0) pm[row_l * 8 + wave] = both;
          }
        }
      }
      lds_barrier();   // partial maxima visible; every wave is done reading the planes of this step's input
      if (tid < 128) *reinterpret_cast<f4*>(cwb + ((last_step ? 0 : s + 1) & 1) * 512 + tid * 4) = cw_next;
      if (!chain_end) {
        // ---- the next step's input planes (its extra segment, if it comes from the colour head, shares the row scale)
        const bool with_rgb = Sn.x_src == 2;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          float mx = pm[row_l * 8];
#pragma unroll
          for (int w = 1; w < 8; ++w) mx = fmaxf(mx, pm[row_l * 8 + w]);
          if (with_rgb) { const f4 g = *reinterpret_cast<const f4*>(rgbs + row_l * 4); mx = fmaxf(mx, ws_absmax4(g)); }
          const float sc = chain_row_scale(mx);
          chain_put16(acc[rt], sc, smem + row_l * CH_ALD + cbase * 2, APLANE);
          if (wave == 0 && half == 0) {
            rs[row_l] = cnr_pow2_rcp(sc); rsf[row_l] = sc;
            const long grow = tile * T + row_l;
            if (Sn.rs_in != nullptr && grow < P) Sn.rs_in[grow] = chain_rs_value(mx, sc);
          }
        }
      }
      if (chain_end) {
        // ---- the 3-wide head on the ReLU output still in registers: fp32 dot products, partial sums per (row, wave) in a fixed order.
        // BEFORE the row stores below: a load issued behind them is waited for with vmcnt(0), i.e. with the whole store queue of the wave
        const ChainFwdHead& H = S.head == 1 ? c.col_head : c.rel_head;
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
          f4 wj[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            wj[q] = f4{0.f, 0.f, 0.f, 0.f};
            if (j < H.n) wj[q] = *reinterpret_cast<const f4*>(H.W + (long)j * H.ldw + cbase + 4 * q);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            float dot = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dot = fmaf(acc[rt][r], wj[r >> 2][r & 3], dot);
            dot = cnr_pair32_sum(dot);
            if (half == 0) hp[((rt * 32 + pt) * 8 + wave) * 4 + j] = dot;
          }
        }
      }
      // ---- the ReLU output rows for the backward pass: 32 x 32 block -> LDS (16 rows at a time) -> 8 rows x 128 contiguous bytes per store
      if (S.save != nullptr && !(CNR_ABLATION(c.dbg) & 1)) {
        const long tile0 = tile * T;
        const int rows_left = (int)((P - tile0) < T ? (P - tile0) : T);
        float* save_tile = S.save + tile0 * S.ld_save + wave * 32;                   // (uniform)
        const int sv_off = (lane >> 3) * S.ld_save + (lane & 7) * 4;                 // this lane's row / 16-byte chunk within an 8-row group
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int hpass = 0; hpass < 2; ++hpass) {
            if ((pt >> 4) == hpass) {
              const int r = pt & 15;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f4 v = {acc[rt][4 * q], acc[rt][4 * q + 1], acc[rt][4 * q + 2], acc[rt][4 * q + 3]};
                *reinterpret_cast<f4*>(tb + r * 128 + (((4 * half + q) ^ (r & 7)) << 4)) = v;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int r = (lane >> 3) + 8 * i, ch = lane & 7;
              const f4 v = *reinterpret_cast<const f4*>(tb + r * 128 + ((ch ^ (r & 7)) << 4));
              const int row0 = rt * 32 + hpass * 16 + 8 * i;                         // first row of this 8-row group within the tile
              if (row0 + (lane >> 3) < rows_left && !(CNR_ABLATION(c.dbg) & 8)) CH_SAVE_STORE(save_tile + row0 * S.ld_save + sv_off, v);   // (read again only by the backward pass)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      if (chain_end) {
        const ChainFwdHead& H = S.head == 1 ? c.col_head : c.rel_head;
        lds_barrier();
        if (tid < T) {
          const int row_l = tid;
          const long grow = tile * T + row_l;
          const long orow = (ridx != nullptr && grow < P) ? (long)ridx[grow] : grow;   // where this row's head outputs go
          float v[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            float sum = hp[(row_l * 8) * 4 + j];
#pragma unroll
            for (int w = 1; w < 8; ++w) sum += hp[(row_l * 8 + w) * 4 + j];
            v[j] = j < H.n ? sum + hbias[j] : 0.0f;
          }
          if (S.head == 1) {
            f4 y = {0.f, 0.f, 0.f, 0.f};
            y.x = c.col_squeeze ? sigmoidf_(v[0]) : v[0];
            if (H.n > 1) y.y = c.col_squeeze ? sigmoidf_(v[1]) : v[1];
            if (H.n > 2) y.z = c.col_squeeze ? sigmoidf_(v[2]) : v[2];
            *reinterpret_cast<f4*>(rgbs + row_l * 4) = y;
            if (grow < P) {
              float* g = c.gcol + orow * 4;
              g[0] = y.x;
              if (H.n > 1) g[1] = y.y;
              if (H.n > 2) g[2] = y.z;
              if (c.rgb_tail != nullptr) {
                const f4 z4 = {0.f, 0.f, 0.f, 0.f};
                f4* t = reinterpret_cast<f4*>(c.rgb_tail + orow * c.ld_tail);
                t[0] = y; t[1] = z4; t[2] = z4; t[3] = z4;
              }
            }
          } else if (grow < P) {
            const f4 rgb = *reinterpret_cast<const f4*>(rgbs + row_l * 4);
            const float rv[3] = {rgb.x, rgb.y, rgb.z};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              if (j < H.n) {
                c.delta[orow * 3 + j] = v[j];
                c.relit[orow * 4 + j] = relight_apply(rv[j], v[j], c.inv_sigmoid);
              }
            }
          }
        }
      }
      lds_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// A wave's 32 x 32 output block (lane = row pt, 16 consecutive columns per half) -> global rows: through the wave's private 2 KB LDS buffer,
// 16 rows at a time (XOR-swizzled 16-byte chunks: conflict-free both ways), stored as 8 rows x 128 contiguous bytes per instruction.
// dst_tile: first row of the tile (uniform) + the wave's first column; row_base: the block's first row within the tile.
// ------------------------------------------------------------------------------------------------
template <bool NT = false>
__device__ __forceinline__ void chain_save_block(const f32x16& a, unsigned char* tb, float* dst_tile, int ld, int row_base, int rows_left, int lane) {
  const int half = lane >> 5, pt = lane & 31;
  const int sv_off = (lane >> 3) * ld + (lane & 7) * 4;
#pragma unroll
  for (int hpass = 0; hpass < 2; ++hpass) {
    if ((pt >> 4) == hpass) {
      const int r = pt & 15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        *reinterpret_cast<f4*>(tb + r * 128 + (((4 * half + q) ^ (r & 7)) << 4)) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (lane >> 3) + 8 * i, ch = lane & 7;
      const f4 v = *reinterpret_cast<const f4*>(tb + r * 128 + ((ch ^ (r & 7)) << 4));
      const int row0 = row_base + hpass * 16 + 8 * i;
      if (row0 + (lane >> 3) < rows_left) { if (NT) CH_SAVE_STORE(dst_tile + row0 * ld + sv_off, v); else *reinterpret_cast<f4*>(dst_tile + row0 * ld + sv_off) = v; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// The SAVING SDF value chain (SdfSaveChain): sdf_value_chain_kernel<RT, 1> of cnr_chain.hip plus the stores the backward pass needs.
// Per hidden layer: z = acc / (row scale x column scale) + bias is stored (through chain_save_block) BEFORE the activation; the layer in front
// of a skip connection stores [z | e] (what the per-layer launches' tail fill writes) and hands [softplus(z) | e] / sqrt(2) to the next layer.
// The top layer's 256 feature rows are one more MFMA step whose output goes straight to the colour network's input buffer.
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(512, 1) void sdf_save_chain_kernel(const SdfSaveChain c) {
  constexpr int T = 32 * RT;
  constexpr int APLANE = T * CH_ALD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rs = reinterpret_cast<float*>(smem + 2 * APLANE);   // [T] 1 / row scale of the current layer input
  float* pm = rs + T;                                        // [T][8] per-wave partial row maxima
  float* pd = pm + T * 8;                                    // [T][8] per-wave partial dot products of the sdf row
  float* cwb = pd + T * 8;                                   // [512] column scales | biases of the step in flight (rewritten behind the row-max barrier: every wave has read its values by then)
  float* wtop = cwb + 512;                                   // [256] sdf row of the top layer
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  unsigned char* tb = reinterpret_cast<unsigned char*>(wtop + 256) + wave * 2048;
  const SdfValueChain& v = c.v;
  const long ntiles = (v.P + T - 1) / T;
  const int nsteps = v.nl + 1;                               // hidden layers + the feature rows of the top layer
  // (the host puts the top layer's feature rows behind the hidden layers, v.lay[v.nl]: one array index for every step -- a choice between
  // two kernel-argument references would make the compiler copy the argument block to scratch)

  f16x8 wr1[4][1], wr2[4][1];
  auto wlane_of = [&](const FusedLayer& L, int lane_) __attribute__((always_inline)) { return L.Wf + (long)wave * (L.K >> 4) * 1024 + lane_ * 8; };
  auto cw_fetch = [&](const FusedLayer& L, int tid) __attribute__((always_inline)) {
    f4 x = {0.f, 0.f, 0.f, 0.f};
    if (tid < 64) x = *reinterpret_cast<const f4*>(L.wsc + tid * 4);
    else if (tid < 128) x = *reinterpret_cast<const f4*>(L.bias + (tid - 64) * 4);
    return x;
  };
  chain_wprime<1>(wr1, wr2, wlane_of(v.lay[0], tid0 & 63), v.lay[0].K >> 4, 0, v.lay[0].K >> 4);
  for (int i = tid0; i < 256; i += 512) wtop[i] = v.wtop[i];
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long tile0 = tile * T;
    const int rows_left = (int)((v.P - tile0) < T ? (v.P - tile0) : T);
    {
      // ---- layer-0 input: E rows -> planes (16 threads per row, 4 columns each)
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      if (tid < 128) *reinterpret_cast<f4*>(cwb + tid * 4) = cw_fetch(v.lay[0], tid);
      const int sc4 = (tid & 15) * 4;
      f4 x[RT];
#pragma unroll
      for (int pass = 0; pass < RT; ++pass) {
        const int row_l = pass * 32 + (tid >> 4);
        const int row_c = row_l < rows_left ? row_l : rows_left - 1;
        x[pass] = f4{0.f, 0.f, 0.f, 0.f};
        if (sc4 < kEmb) x[pass] = *reinterpret_cast<const f4*>(v.E + (tile0 + row_c) * kEmb + sc4);
      }
#pragma unroll
      for (int pass = 0; pass < RT; ++pass) {
        const int row_l = pass * 32 + (tid >> 4);
        float mx = ws_absmax4(x[pass]);
        mx = cnr_max16(mx);
        const float sc = chain_row_scale(mx);
        if (sc4 < kEmb) chain_put4(x[pass], sc, smem + row_l * CH_ALD + sc4 * 2, APLANE);
        if ((tid & 15) == 0) rs[row_l] = cnr_pow2_rcp(sc);
      }
      lds_barrier();
    }
    for (int l = 0; l < nsteps; ++l) {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, half = lane >> 5, pt = lane & 31;
      const int cbase = wave * 32 + 16 * half;
      const bool is_top = l == v.nl, last_hidden = l + 1 == v.nl;
      const FusedLayer& L = v.lay[l];
      const FusedLayer& Ln = v.lay[is_top ? 0 : l + 1];
      const int nkb = L.K >> 4;
      const f4 cw_next = cw_fetch(Ln, tid);
      f32x16 acc1[1][RT];
      f32x16 (&acc)[RT] = acc1[0];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
      const unsigned char* Ab = smem + pt * CH_ALD + half * 16;
      if (nkb == 16) chain_mfma_blocks<RT, 1, 16>(acc1, wr1, wr2, Ab, APLANE, wlane_of(L, lane), 16, 0);
      else chain_mfma_blocks<RT, 1, 3>(acc1, wr1, wr2, Ab, APLANE, wlane_of(L, lane), 3, 0);
      chain_wprime<1>(wr1, wr2, wlane_of(Ln, lane), Ln.K >> 4, 0, Ln.K >> 4);

      const float* cw = cwb;
      const bool next_skip = !is_top && !last_hidden && ((v.skip_mask >> (l + 1)) & 1);
      const bool ragged = !is_top && L.N < 256;             // wave-uniform: only the layer in front of a skip connection
      float* out_tile = (is_top ? c.feat + tile0 * c.ld_feat : c.Z[l] + tile0 * c.ldz) + wave * 32;
      const int out_ld = is_top ? c.ld_feat : c.ldz;
      float rmax[RT], rdot[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { rmax[rt] = 0.0f; rdot[rt] = 0.0f; }
      {
        f4 wsc4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wsc4[q] = *reinterpret_cast<const f4*>(cw + cbase + 4 * q);
          b4[q] = *reinterpret_cast<const f4*>(cw + 256 + cbase + 4 * q);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          const f2 rsc = pk_splat(rs[row_l]);
          // z (or, for the top layer, the feature row): acc / (row scale * column scale) + bias
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f2 w = {wsc4[q >> 1][(2 * q) & 3], wsc4[q >> 1][(2 * q + 1) & 3]};
            const f2 b = {b4[q >> 1][(2 * q) & 3], b4[q >> 1][(2 * q + 1) & 3]};
            f2 a = {acc[rt][2 * q], acc[rt][2 * q + 1]};
            a = pk_fma(a, rsc * w, b);
            acc[rt][2 * q] = a.x; acc[rt][2 * q + 1] = a.y;
          }
          const bool tail_lane = ragged && cbase + 16 > L.N;  // owns columns >= N: they carry e (fields.py:86-87)
          if (tail_lane) {
            const int row_c = row_l < rows_left ? row_l : rows_left - 1;
            const float* erow = v.E + (tile0 + row_c) * kEmb;
            float ev[16];   // (all 16 loads first: consumed one by one the compiler waits for each before it issues the next)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int ei = cbase + r - L.N; ev[r] = erow[ei < 0 ? 0 : (ei < kEmb ? ei : kEmb - 1)]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ei = cbase + r - L.N;
              acc[rt][r] = ei < 0 ? acc[rt][r] : ((next_skip && ei < v.emb) ? ev[r] : 0.0f);
            }
          }
          if (is_top) chain_save_block<false>(acc[rt], tb, out_tile, out_ld, rt * 32, rows_left, lane);   // (the feature rows: the colour network reads them next)
          else chain_save_block<true>(acc[rt], tb, out_tile, out_ld, rt * 32, rows_left, lane);           // (pre-activations: read again only by the gradient chain / backward pass)
          if (!is_top) {
            // a = softplus(z) (the tail columns keep e), / sqrt(2) in front of a skip layer ; row max ; partial dot with the sdf row
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              f2 z = {acc[rt][2 * q], acc[rt][2 * q + 1]};
              f2 a = softplus100_pk(z);
              if (tail_lane) { a.x = cbase + 2 * q >= L.N ? z.x : a.x; a.y = cbase + 2 * q + 1 >= L.N ? z.y : a.y; }
              if (next_skip) a = a * pk_splat(kInvSqrt2);
              acc[rt][2 * q] = a.x; acc[rt][2 * q + 1] = a.y;
            }
            float mx = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(fabsf(acc[rt][r]), fabsf(acc[rt][r + 1])), mx);
            rmax[rt] = mx;
            if (last_hidden) {
              float dot = 0.0f;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f4 wt = *reinterpret_cast<const f4*>(wtop + cbase + 4 * q);
                dot = fmaf(acc[rt][4 * q], wt.x, dot); dot = fmaf(acc[rt][4 * q + 1], wt.y, dot);
                dot = fmaf(acc[rt][4 * q + 2], wt.z, dot); dot = fmaf(acc[rt][4 * q + 3], wt.w, dot);
              }
              rdot[rt] = dot;
            }
          }
        }
      }
      if (!is_top) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float o = cnr_pair32_max(rmax[rt]);
          if (half == 0) pm[(rt * 32 + pt) * 8 + wave] = o;
          if (last_hidden) { const float od = cnr_pair32_sum(rdot[rt]); if (half == 0) pd[(rt * 32 + pt) * 8 + wave] = od; }
        }
      }
      lds_barrier();   // partial maxima / dots visible; every wave is done reading the planes of this step's input
      if (tid < 128) *reinterpret_cast<f4*>(cwb + tid * 4) = cw_next;
      if (!is_top) {
        float* rso = c.rs[l + 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          float mx = pm[row_l * 8];
#pragma unroll
          for (int w = 1; w < 8; ++w) mx = fmaxf(mx, pm[row_l * 8 + w]);
          const float sc = chain_row_scale(mx);
          chain_put16(acc[rt], sc, smem + row_l * CH_ALD + cbase * 2, APLANE);
          if (wave == 0 && half == 0) {
            rs[row_l] = cnr_pow2_rcp(sc);
            if (rso != nullptr && row_l < rows_left) rso[tile0 + row_l] = chain_rs_value(mx, sc);
          }
        }
        if (last_hidden && tid < T) {
          // sdf = (softplus(z_top-1) . w_sdf + b_sdf) * top_scale: the per-wave partial sums in a fixed order
          float sum = pd[tid * 8];
#pragma unroll
          for (int w = 1; w < 8; ++w) sum += pd[tid * 8 + w];
          if (tid < rows_left) v.sdf_out[tile0 + tid] = (sum + v.btop[0]) * v.top_scale;
        }
      }
      lds_barrier();
    }
  }
}

template <int RT>
static void launch_sdf_save_chain(const SdfSaveChain& c, cnr_stream s) {
  constexpr int T = 32 * RT;
  const size_t lds = (size_t)2 * T * CH_ALD + (size_t)T * 17 * sizeof(float) + (size_t)(512 + 256) * sizeof(float) + (size_t)8 * 2048;
  static_assert((size_t)2 * T * CH_ALD + (size_t)T * 17 * sizeof(float) + (size_t)(512 + 256) * sizeof(float) + (size_t)8 * 2048 <= 160 * 1024, "LDS budget");
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_save_chain_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long ntiles = (c.v.P + T - 1) / T;
  const int wgs_env = debug_flags().chain_wgs;
  const long wgs = wgs_env > 0 ? wgs_env : 256;
  const unsigned grid = (unsigned)(ntiles < wgs ? ntiles : wgs);
  double macs = 256.0 * 256.0, bytes = (double)c.v.P * ((kEmb + 1) * 4.0 + 1024.0);
  for (int l = 0; l < c.v.nl; ++l) { macs += (double)c.v.lay[l].K * 256.0; bytes += (double)c.v.P * (1024.0 + (c.rs[l + 1] ? 4.0 : 0.0)); }
  TimingScope ts_("chain_sdf_fwd", 3, RT * 10 + 1, c.v.P, (int)(macs / 256.0), 256, 1, s, bytes);
  SdfSaveChain cc = c;
  cc.v.lay[c.v.nl] = c.top;
  hipLaunchKernelGGL((sdf_save_chain_kernel<RT>), dim3(grid), dim3(512), lds, s, cc);
}

bool be_sdf_save_chain(const SdfSaveChain& c, cnr_stream s) {
  const bool off = debug_flags().no_fused || debug_flags().no_chain_sdf;   // debugging aids: per-layer launches
  const SdfValueChain& v = c.v;
  if (off || v.P <= 0 || v.nl < 1 || v.nl >= kMaxLayers) return false;
  for (int l = 0; l < v.nl; ++l) {
    if ((v.lay[l].K != 256 && !(l == 0 && v.lay[l].K == 48)) || v.lay[l].N > 256 || v.lay[l].N < 1 || !v.lay[l].Wf || !c.Z[l]) return false;
    if (l + 1 < v.nl && v.lay[l].N < 256 && !((v.skip_mask >> (l + 1)) & 1)) return false;   // a narrow layer only in front of a skip connection
  }
  if (v.lay[v.nl - 1].N != 256 || c.top.K != 256 || c.top.N != 256 || !c.top.Wf || !c.feat || (c.ld_feat & 3) || (c.ldz & 3) || c.ldz < 256) return false;
  const int force = debug_flags().chain_fwd_rt;
  const int rt = force ? force : (v.P >= 256L * 128 ? 4 : (v.P >= 256L * 64 ? 2 : 1));
  if (rt == 4) launch_sdf_save_chain<4>(c, s);
  else if (rt == 2) launch_sdf_save_chain<2>(c, s);
  else launch_sdf_save_chain<1>(c, s);
  CNR_LAUNCH_CHECK("chain_sdf_fwd");
  return true;
}

// ------------------------------------------------------------------------------------------------
// Global rows -> a wave's 32 x 32 block in the chain layout (lane = row pt, 16 consecutive columns per half): the reverse of chain_save_block.
// The 16-byte pieces are requested first (chain_load_issue: 8 rows x 128 contiguous bytes per instruction) and pass through the wave's
// LDS buffer later (chain_load_finish), so that the memory latency hides behind whatever the caller does in between.
// ------------------------------------------------------------------------------------------------
struct ChainRaw { f4 v[2][2]; };   // [16-row pass][8-row group]
__device__ __forceinline__ ChainRaw chain_load_issue(const float* src_tile, int ld, int row_base, int rows_left, int lane) {
  ChainRaw r;
#pragma unroll
  for (int hpass = 0; hpass < 2; ++hpass)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = row_base + hpass * 16 + 8 * i + (lane >> 3);
      if (row >= rows_left) row = rows_left - 1;             // rows past the end read the last row (their results are never stored)
      r.v[hpass][i] = *reinterpret_cast<const f4*>(src_tile + row * ld + (lane & 7) * 4);
    }
  return r;
}
__device__ __forceinline__ void chain_load_finish(const ChainRaw& raw, unsigned char* tb, float (&z)[16], int lane) {
  const int half = lane >> 5, pt = lane & 31;
#pragma unroll
  for (int hpass = 0; hpass < 2; ++hpass) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (lane >> 3) + 8 * i, ch = lane & 7;
      *reinterpret_cast<f4*>(tb + r * 128 + ((ch ^ (r & 7)) << 4)) = raw.v[hpass][i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((pt >> 4) == hpass) {
      const int r = pt & 15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 v = *reinterpret_cast<const f4*>(tb + r * 128 + (((4 * half + q) ^ (r & 7)) << 4));
        z[4 * q] = v.x; z[4 * q + 1] = v.y; z[4 * q + 2] = v.z; z[4 * q + 3] = v.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// The analytic gradient chain (SdfGradChain) on a 32 * RT point tile.  Step l: planes = u_l (hi / lo of the exactly scaled row);
// v_{l-1} = W_l^T u_l by the transposed MFMA product; the epilogue undoes the scales, stores v_{l-1} (chain_save_block), fetches z_{l-1} in the
// lane layout (chain_load_*), forms u_{l-1} = softplus'(z_{l-1}) * v_{l-1}, its row scale, and the next planes.
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(512, 1) void sdf_grad_chain_kernel(const SdfGradChain c) {
  constexpr int T = 32 * RT;
  constexpr int APLANE = T * CH_ALD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rs = reinterpret_cast<float*>(smem + 2 * APLANE);   // [T] 1 / row scale of the current step's input
  float* pm = rs + T;                                        // [T][8] per-wave partial row maxima
  float* cwb = pm + T * 8;                                   // [256] column scales of the step in flight
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  unsigned char* tb = reinterpret_cast<unsigned char*>(cwb + 256) + wave * 2048;
  const long ntiles = (c.P + T - 1) / T;
  const int L = c.nl;

  f16x8 wr1[4][1], wr2[4][1];
  auto wlane_of = [&](const FusedLayer& Q, int lane_) __attribute__((always_inline)) { return Q.Wf + (long)wave * (Q.K >> 4) * 1024 + lane_ * 8; };
  chain_wprime<1>(wr1, wr2, wlane_of(c.lay[L - 1], tid0 & 63), c.lay[L - 1].K >> 4, 0, c.lay[L - 1].K >> 4);
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long tile0 = tile * T;
    const int rows_left = (int)((c.P - tile0) < T ? (c.P - tile0) : T);
    {
      // ---- the chain starts: u_{L-1} = softplus'(z_{L-1}) * (vrow * vscale), rows from HBM (16 threads per row, 4 columns each per 64)
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      if (tid < 64) *reinterpret_cast<f4*>(cwb + tid * 4) = *reinterpret_cast<const f4*>(c.lay[L - 1].wsc + tid * 4);
      const int sc4 = (tid & 15) * 4;
      const float* zt = c.Z[L - 1] + tile0 * c.ldz;
      f4 w4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w4[j] = *reinterpret_cast<const f4*>(c.vrow + 64 * j + sc4);
      f4 zv[RT][4];
#pragma unroll
      for (int pass = 0; pass < RT; ++pass) {
        const int row_l = pass * 32 + (tid >> 4);
        const int row_c = row_l < rows_left ? row_l : rows_left - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) zv[pass][j] = *reinterpret_cast<const f4*>(zt + row_c * c.ldz + 64 * j + sc4);
      }
#pragma unroll
      for (int pass = 0; pass < RT; ++pass) {
        const int row_l = pass * 32 + (tid >> 4);
        float mx = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f4& u = zv[pass][j];
          u.x = softplus100_d1(u.x) * w4[j].x * c.vscale; u.y = softplus100_d1(u.y) * w4[j].y * c.vscale;
          u.z = softplus100_d1(u.z) * w4[j].z * c.vscale; u.w = softplus100_d1(u.w) * w4[j].w * c.vscale;
          mx = fmaxf(mx, ws_absmax4(u));
        }
        mx = cnr_max16(mx);
        const float sc = chain_row_scale(mx);
        unsigned char* dst = smem + row_l * CH_ALD + sc4 * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) chain_put4(zv[pass][j], sc, dst + 128 * j, APLANE);
        if ((tid & 15) == 0) {
          rs[row_l] = cnr_pow2_rcp(sc);
          if (c.rs[L - 1] != nullptr && row_l < rows_left) c.rs[L - 1][tile0 + row_l] = chain_rs_value(mx, sc);
        }
      }
      lds_barrier();
    }
    for (int l = L - 1; l >= 0; --l) {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, half = lane >> 5, pt = lane & 31;
      const int cbase = wave * 32 + 16 * half;
      const FusedLayer& Q = c.lay[l];
      const FusedLayer& Qn = c.lay[l == 0 ? L - 1 : l - 1];
      const int nkb = Q.K >> 4;
      f4 cw_next = {0.f, 0.f, 0.f, 0.f};
      if (tid < 64) cw_next = *reinterpret_cast<const f4*>(Qn.wsc + tid * 4);
      f32x16 acc1[1][RT];
      f32x16 (&acc)[RT] = acc1[0];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
      const unsigned char* Ab = smem + pt * CH_ALD + half * 16;
      // z_{l-1} of this wave's first row blocks is requested BEFORE the MFMA phase (its latency hides behind the matrix work), the rest at the
      // start of the epilogue (hidden behind the first blocks' stores and sigmoid math): 32 + 32 registers in flight instead of 64
      constexpr int RG = RT < 2 ? RT : 2;
      ChainRaw zraw0[RG];
      if (l > 0) {
        const float* zt = c.Z[l - 1] + tile0 * c.ldz + wave * 32;
#pragma unroll
        for (int k = 0; k < RG; ++k) zraw0[k] = chain_load_issue(zt, c.ldz, k * 32, rows_left, lane);
      }
      if (nkb == 16) chain_mfma_blocks<RT, 1, 16>(acc1, wr1, wr2, Ab, APLANE, wlane_of(Q, lane), 16, 0);
      else chain_mfma_blocks<RT, 1, 14>(acc1, wr1, wr2, Ab, APLANE, wlane_of(Q, lane), 14, 0);
      chain_wprime<1>(wr1, wr2, wlane_of(Qn, lane), Qn.K >> 4, 0, Qn.K >> 4);

      // ---- epilogue
      const bool skip = (c.skip_mask >> l) & 1;
      const float oscale = skip ? kInvSqrt2 : 1.0f;
      const int nv = l > 0 ? c.n_out[l] : 0;                 // output columns that are v_{l-1} (the hidden width of the layer below)
      const int nlive = Q.N;                                 // live output columns (v part + embedding part)
      float rmax[RT];
      {
        f4 wsc4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wsc4[q] = *reinterpret_cast<const f4*>(cwb + cbase + 4 * q);
        ChainRaw zraw1[RG];
        if (l > 0 && RT > RG) {
          const float* zt = c.Z[l - 1] + tile0 * c.ldz + wave * 32;
#pragma unroll
          for (int k = 0; k < RG; ++k) zraw1[k] = chain_load_issue(zt, c.ldz, (RG + k) * 32, rows_left, lane);
        }
#pragma unroll
        for (int g0 = 0; g0 < RT; g0 += RG) {
          ChainRaw (&zraw)[RG] = g0 == 0 ? zraw0 : zraw1;
#pragma unroll
          for (int k = 0; k < RG; ++k) {
            const int rt = g0 + k;
            const int row_l = rt * 32 + pt;
            const float rsc = rs[row_l];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float o = (acc[rt][r] * (rsc * wsc4[r >> 2][r & 3])) * oscale;
              acc[rt][r] = cbase + r < nlive ? o : 0.0f;     // (columns without weights: their scales are not even defined)
            }
            if (l == 0) {
              // embedding cotangent of layer 0: [P][kEmb], kEmb = 48 columns (waves 0 and 1 hold them)
              if (wave == 0) chain_save_block(acc[rt], tb, c.ce0 + tile0 * kEmb, kEmb, rt * 32, rows_left, lane);
              else if (wave == 1) {
                // columns 32..47: the left half of this wave's block (lanes of half 0), 64 bytes per row
                if (half == 0 && row_l < rows_left) {
                  float* d = c.ce0 + (tile0 + row_l) * kEmb + 32;
#pragma unroll
                  for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(d + 4 * q) = f4{acc[rt][4 * q], acc[rt][4 * q + 1], acc[rt][4 * q + 2], acc[rt][4 * q + 3]};
                }
              }
            } else {
              if (skip && cbase + 16 > nv) {
                // embedding part of a skip layer's output -> ces (scalar stores: 39 floats per row), then zero in the row that becomes v_{l-1}
                if (row_l < rows_left) {
                  float* d = c.ces + (tile0 + row_l) * kEmb + c.ces_off - nv;
#pragma unroll
                  for (int r = 0; r < 16; ++r) if (cbase + r >= nv && cbase + r < nlive) d[cbase + r] = acc[rt][r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][r] = cbase + r < nv ? acc[rt][r] : 0.0f;
              }
              chain_save_block(acc[rt], tb, c.V[l - 1] + tile0 * c.ldz + wave * 32, c.ldz, rt * 32, rows_left, lane);
            }
          }
          if (l > 0) {
            // u_{l-1} = softplus'(z_{l-1}) * v_{l-1} ; row max
#pragma unroll
            for (int k = 0; k < RG; ++k) {
              const int rt = g0 + k;
              float z[16];
              chain_load_finish(zraw[k], tb, z, lane);
              float mx = 0.0f;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float u = softplus100_d1(z[r]) * acc[rt][r];
                acc[rt][r] = cbase + r < nv ? u : 0.0f;      // (the tail columns of z below a skip connection hold e, not a pre-activation)
                mx = fmaxf(mx, fabsf(acc[rt][r]));
              }
              rmax[rt] = mx;
            }
          }
        }
        if (l > 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const float o = cnr_pair32_max(rmax[rt]);
            if (half == 0) pm[(rt * 32 + pt) * 8 + wave] = o;
          }
        }
      }
      lds_barrier();   // partial maxima visible; every wave is done reading the planes of this step's input and its column scales
      if (tid < 64) *reinterpret_cast<f4*>(cwb + tid * 4) = cw_next;
      if (l > 0) {
        float* rso = c.rs[l - 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row_l = rt * 32 + pt;
          float mx = pm[row_l * 8];
#pragma unroll
          for (int w = 1; w < 8; ++w) mx = fmaxf(mx, pm[row_l * 8 + w]);
          const float sc = chain_row_scale(mx);
          chain_put16(acc[rt], sc, smem + row_l * CH_ALD + cbase * 2, APLANE);
          if (wave == 0 && half == 0) {
            rs[row_l] = cnr_pow2_rcp(sc);
            if (rso != nullptr && row_l < rows_left) rso[tile0 + row_l] = chain_rs_value(mx, sc);
          }
        }
      }
      lds_barrier();
    }
  }
}

template <int RT>
static void launch_sdf_grad_chain(const SdfGradChain& c, cnr_stream s) {
  constexpr int T = 32 * RT;
  const size_t lds = (size_t)2 * T * CH_ALD + (size_t)T * 9 * sizeof(float) + (size_t)256 * sizeof(float) + (size_t)8 * 2048;
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sdf_grad_chain_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long ntiles = (c.P + T - 1) / T;
  const int wgs_env = debug_flags().chain_wgs;
  const long wgs = wgs_env > 0 ? wgs_env : 256;
  const unsigned grid = (unsigned)(ntiles < wgs ? ntiles : wgs);
  double macs = 0.0, bytes = (double)c.P * (2.0 * kEmb * 4.0);
  for (int l = 0; l < c.nl; ++l) { macs += (double)c.lay[l].K * 256.0; bytes += (double)c.P * (1024.0 + (l > 0 ? 1024.0 : 0.0) + (c.rs[l] ? 4.0 : 0.0)); }
  TimingScope ts_("chain_sdf_grad", 3, RT * 10 + 1, c.P, (int)(macs / 256.0), 256, 1, s, bytes);
  hipLaunchKernelGGL((sdf_grad_chain_kernel<RT>), dim3(grid), dim3(512), lds, s, c);
}

bool be_sdf_grad_chain(const SdfGradChain& c, cnr_stream s) {
  // OFF by default: measured slower than the 8 per-layer launches at every batch size (4096 rays: 3.6 ms against 2.3 ms; 512 rays: 0.50 against
  // 0.35 ms) -- each step's epilogue has to fetch z_{l-1} (1 KB per point, through the LDS transposition in the other direction) and run a
  // sigmoid per element before the next MFMA phase can start, and none of that overlaps the matrix work.  Kept as a
  // tested alternative (CNR_CHAIN_GRAD=1; tests/test_hip_parity.py runs the strict gate on it).
  const bool on = debug_flags().chain_grad && !debug_flags().no_fused;
  if (!on || c.P <= 0 || c.nl < 2 || c.nl > kMaxLayers || !c.vrow || !c.ce0 || (c.ldz & 3) || c.ldz < 256) return false;
  for (int l = 0; l < c.nl; ++l) {
    const FusedLayer& Q = c.lay[l];
    if (!Q.Wf || !Q.wsc || (Q.K != 256 && Q.K != 224) || Q.N < 1 || Q.N > 256 || !c.Z[l]) return false;
    if (l > 0 && (!c.V[l - 1] || c.n_out[l] < 1 || c.n_out[l] > Q.N)) return false;
    if (((c.skip_mask >> l) & 1) && (!c.ces || l == 0)) return false;
    if (l > 0 && !((c.skip_mask >> l) & 1) && c.n_out[l] != Q.N) return false;
    if (l > 0 && c.lay[l - 1].K < c.n_out[l]) return false;   // the planes of step l - 1 hold the v part
  }
  if (c.lay[0].N > kEmb) return false;
  const int force = debug_flags().chain_fwd_rt;
  const int rt = force ? force : (c.P >= 256L * 128 ? 4 : (c.P >= 256L * 64 ? 2 : 1));
  if (rt == 4) launch_sdf_grad_chain<4>(c, s);
  else if (rt == 2) launch_sdf_grad_chain<2>(c, s);
  else launch_sdf_grad_chain<1>(c, s);
  CNR_LAUNCH_CHECK("chain_sdf_grad");
  return true;
}

template <int RT>
static void launch_relu_chain_fwd(const ReluChainFwd& c, cnr_stream s) {
  constexpr int T = 32 * RT;
  const size_t lds = (size_t)2 * T * CH_ALD + (size_t)T * (2 + 8 + 4) * sizeof(float) + (size_t)1024 * sizeof(float) + (size_t)8 * 2048;
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&relu_chain_fwd_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long ntiles = (c.P + T - 1) / T;
  const int wgs_env = debug_flags().chain_wgs;
  const long wgs = wgs_env > 0 ? wgs_env : 256;   // persistent, one workgroup per CU
  const unsigned grid = (unsigned)(ntiles < wgs ? ntiles : wgs);
  double macs = 0.0, bytes = 0.0;
  for (int i = 0; i < c.nsteps; ++i) {
    const ChainFwdStep& S = c.st[i];
    macs += 16.0 * (S.nkb_main + S.nkb_x) * 256.0;
    if (S.in) bytes += (double)c.P * 4.0 * 16.0 * (S.nkb_main + (S.x_src == 1 ? S.nkb_x : 0));
    if (S.save) bytes += (double)c.P * 1024.0;
    if (S.rs_in) bytes += (double)c.P * 4.0;
    if (S.head) bytes += (double)c.P * (S.head == 1 ? (c.rgb_tail ? 80.0 : 16.0) : 28.0);
  }
  TimingScope ts_("chain_fwd", 3, RT * 10 + 1, c.P, (int)(macs / 256.0), 256, 1, s, bytes);
  const int dbg = debug_flags().chain_fwd_dbg;
  ReluChainFwd cc = c;
  cc.dbg = dbg;
  hipLaunchKernelGGL((relu_chain_fwd_kernel<RT>), dim3(grid), dim3(512), lds, s, cc);
}

bool be_relu_chain_fwd_enabled() {
  const bool off = debug_flags().no_fused || debug_flags().no_chain_fwd;   // debugging aids: per-layer launches
  return !off;
}

bool be_relu_chain_fwd(const ReluChainFwd& c, cnr_stream s) {
  const bool off = !be_relu_chain_fwd_enabled();
  if ((c.row_idx != nullptr) != (c.P_dev != nullptr)) return false;
  if (off || c.P <= 0 || c.nsteps < 1 || c.nsteps > kChainSteps) return false;
  for (int i = 0; i < c.nsteps; ++i) {
    const ChainFwdStep& S = c.st[i];
    const bool main_ok = S.in != nullptr ? (S.nkb_main == 16 || (S.nkb_main >= 1 && S.nkb_main <= 3)) : S.nkb_main == 16;
    if (!S.Wf || !S.wsc || !S.bias || !main_ok || S.nkb_x < 0 || S.nkb_x > 3 || S.nkb_main + S.nkb_x > S.nkb_w) return false;
    if (S.nkb_x > 0 && S.x_src != 1 && S.x_src != 2) return false;
    if (S.x_src == 1 && (S.in == nullptr || (S.ld_in & 3) != 0 || S.ld_in < 16 * (S.nkb_main + S.nkb_x))) return false;
    if (S.in != nullptr && ((S.ld_in & 3) != 0 || S.ld_in < 16 * S.nkb_main)) return false;
    if (S.x_src == 2 && S.nkb_x != 1) return false;
    if (S.save != nullptr && (S.ld_save & 3) != 0) return false;
    if (c.row_idx != nullptr && (S.save != nullptr || S.rs_in != nullptr)) return false;   // a compacted run keeps nothing for a backward pass
    if (i == 0 && S.in == nullptr) return false;
    if (i > 0 && c.st[i - 1].head != 0 && S.in == nullptr) return false;   // a chain starts from global rows
    if (S.head == 1 && (!c.col_head.W || c.col_head.n < 1 || c.col_head.n > 3 || (c.col_head.ldw & 3) || !c.gcol || (c.rgb_tail && (c.ld_tail & 3)))) return false;
    if (S.head == 2 && (!c.rel_head.W || c.rel_head.n < 1 || c.rel_head.n > 3 || (c.rel_head.ldw & 3) || !c.delta || !c.relit)) return false;
  }
  if (c.st[c.nsteps - 1].head == 0) return false;
  const int force = debug_flags().chain_fwd_rt;   // tuning aid: 4, 2, 1
  const int rt = force ? force : (c.P >= 256L * 128 ? 4 : (c.P >= 256L * 64 ? 2 : 1));
  if (rt == 4) launch_relu_chain_fwd<4>(c, s);
  else if (rt == 2) launch_relu_chain_fwd<2>(c, s);
  else launch_relu_chain_fwd<1>(c, s);
  CNR_LAUNCH_CHECK("chain_fwd");
  return true;
}

}  // namespace cnr
