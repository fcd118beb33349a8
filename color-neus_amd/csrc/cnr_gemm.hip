// Layer GEMM dispatch for gfx950 (MI355X / CDNA4): the fully-connected layers of the SDF / colour / relight stacks (forward,
// input-gradient chain, second-order sweep, backward).
//
//   be_layer_gemm            splits a layer into <= 256-column launches and picks the kernel:
//     layer_gemm_ws_kernel   (cnr_gemm_ws.h, instantiated in cnr_gemm_ws_a.hip / _b.hip) weight-stationary split-f16 kernel for
//                            >= 96 output columns and K <= 272: the HBM-bound streaming kernel that carries 2/3 of a training step;
//     layer_gemm_kernel<NT>  (cnr_gemm_fp32.h) FP32-MFMA 128-point tiles: narrow outputs (sdf / rgb / embedding cotangents, view
//                            and epilogue kinds pinned at compile time for the combinations of the plan) and the fallback when
//                            the weight-stationary kernel is switched off (CNR_DISABLE_WS) or not applicable.
//   row_scale_kernel         slow-path producer of LayerGemm::rs_out for launches that bypass the weight-stationary kernel.
// The weight-gradient GEMMs live in cnr_gemm_dw.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cnr_backend.h"
#include "cnr_hip_util.h"
#include "cnr_gemm_int.h"
#include "cnr_gemm_fp32.h"

namespace cnr {

static void dispatch_layer_gemm(const LayerGemm& g, int nt, cnr_stream s) {
  // the narrow launches of the render plan (sdf / rgb / embedding-cotangent columns) with their kinds pinned at compile time
  const int vk = g.A.kind, ek = g.E.kind;
#define LG_CASE(V_, E_)                                                              \
  if (vk == V_ && ek == E_ && nt <= 2) {                                              \
    if (nt == 1) launch_layer_gemm<1, V_, E_>(g, s); else launch_layer_gemm<2, V_, E_>(g, s); \
    return;                                                                           \
  }
  LG_CASE(VK_SOFTPLUS, EK_STORE) LG_CASE(VK_SIGMUL, EK_STORE) LG_CASE(VK_DIRECT, EK_STORE) LG_CASE(VK_DIRECT, EK_SIGMOID)
  LG_CASE(VK_DIRECT, EK_RELIGHT_TOP) LG_CASE(VK_DIRECT, EK_RELU_MASK) LG_CASE(VK_DIRECT, EK_SPLIT)
#undef LG_CASE
  switch (nt) {
    case 1: launch_layer_gemm<1>(g, s); break;
    case 2: launch_layer_gemm<2>(g, s); break;
    case 3: launch_layer_gemm<3>(g, s); break;
    case 4: launch_layer_gemm<4>(g, s); break;
    case 5: case 6: case 7: case 8: launch_layer_gemm_wide(g, nt, s); break;   // cnr_gemm_wide.hip
    default:
      if (g_first_error == hipSuccess) { g_first_error = hipErrorInvalidValue; g_first_error_where = "layer_gemm: tile count"; }
      return;
  }
}

// Row scales for a launch that does not go through the weight-stationary kernel (debug switches, unusual shapes): same
// definition as WS_PUT_TILE, one thread per row.  Slow path, kept only so that LayerGemm::rs_out is always honoured.
__global__ void row_scale_kernel(const LayerGemm g) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long Pn = g.P_dev ? (long)*g.P_dev : g.P;
  if (row >= Pn) return;
  const int kpad = ((g.K + 15) >> 4) * 16;
  float mx = 0.0f;
  for (int c = 0; c < kpad; c += 4) mx = fmaxf(mx, ws_absmax4(view_eval4(g.A, row, c)));
  float sc = mx == 0.0f ? 0.0f : __builtin_nanf("");   // 0: all-zero row, NaN: non-finite row
  if (mx > 0.0f && mx < 3.0e38f) { int e_; (void)frexpf(mx, &e_); if (e_ < -100) e_ = -100; sc = ldexpf(1.0f, 14 - e_); }
  g.rs_out[row] = sc;
}

void be_layer_gemm(const LayerGemm& g, cnr_stream s) {
  const bool ws_off = debug_flags().disable_ws;   // debugging aid: force the FP32-MFMA kernel everywhere
  const int ws_kinds = debug_flags().ws_kinds;    // tuning aid: bit mask of epilogue kinds
  int ncols = g.N;
  if (g.E.tail_src && g.E.n_out + g.E.tail_n > ncols) ncols = g.E.n_out + g.E.tail_n;   // tail-fill columns need a tile too
  // layers wider than 256 columns (257 = sdf + features, 259/262 = cotangents of concatenated inputs) run as a 256-wide launch
  // at two workgroups per CU plus a narrow launch for the remaining columns (cheaper than one 288-wide tile at one workgroup per CU)
  for (int c0 = g.first_col; c0 < ncols; c0 += 256) {
    LayerGemm part = g;
    part.col0 = c0;
    const int w = ncols - c0 < 256 ? ncols - c0 : 256;
    part.N = w;   // (only used for the timing record; the kernel takes its extents from K, P and the epilogue)
    const bool k_ok = g.K <= 256 || (g.K <= 272 && ws_k17_supported(g));
    part.rs_out = c0 == 0 ? g.rs_out : nullptr;   // one launch per operand writes the row scales
    const int ws_minw = debug_flags().ws_minw;   // tuning knob (96): narrower launches take the FP32-MFMA kernel
    const bool use_ws = !ws_off && ((ws_kinds >> g.E.kind) & 1) && g.Wp != nullptr && g.wscale != nullptr && w >= ws_minw && k_ok && (g.A.lda & 3) == 0;
    // the row dot is formed by the weight-stationary kernels while they stage the rows (K <= 256, first column range); otherwise by a
    // one-column launch of its own
    const bool dot_native = use_ws && c0 == 0 && g.K <= 256 && g.E.kind == EK_SDF_TOP;
    if (g.dot_w && !(dot_native)) part.dot_w = nullptr;
    if (g.dot_w && c0 == 0 && !dot_native) {
      LayerGemm t;
      t.A = g.A; t.W = g.dot_w; t.ldw = g.ldw; t.N = 1; t.K = g.K; t.P = g.P; t.P_dev = g.P_dev;
      t.E.kind = EK_STORE; t.E.n_out = 1; t.E.bias = g.dot_bias; t.E.scale = g.dot_scale; t.E.o1 = g.dot_out; t.E.ld1 = 1;
      dispatch_layer_gemm(t, 1, s);
    }
    if (c0 != 0) part.dot_w = nullptr;
    if (use_ws) launch_layer_gemm_ws(part, g.w_rows > 0 ? g.w_rows : round_up(g.N, 32), s);
    else {
      dispatch_layer_gemm(part, (w + 31) / 32, s);
      if (part.rs_out && g.P > 0) hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((g.P + 255) / 256)), dim3(256), 0, s, part);
    }
  }
  CNR_LAUNCH_CHECK("layer_gemm");
}

}  // namespace cnr
