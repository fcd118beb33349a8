// Debug / fallback switches of the render library: ONE place reads the environment (debug_flags(), cnr_plan.cpp), once per process, on first
// use; every other file reads the parsed struct.  Nothing here changes a result beyond what the selected kernel form's own round-off
// changes -- except the ablation words of the CNR_TUNING build, which are documented as wrong-result timing aids.
//
// PRODUCT build (make hip): the switches below that the GPU parity tests exercise (tests/test_hip_parity.py::test_fallback_kernels_keep_parity,
// tests/test_edge_batches.py, tests/test_forward_only.py) select a fallback FORM of a launch; the tuning words are compile-time constants, so
// their branches fold away and the ablation code is not in the shipped kernels.
// TUNING build (make hip-tuning -> tools/_build/libcolorneus_hip_tuning.so, -DCNR_TUNING; loaded by the measurement tools through CNR_LIB, never by the
// product path): the tuning words are read from the environment as well.
//
//   environment variable     member            meaning
//   CNR_NO_FUSED             no_fused          per-layer launches instead of every chain-fused kernel (value chain, saving chains, gradient chain)
//   CNR_NO_CHAIN_FWD         no_chain_fwd      per-layer launches instead of the chain-fused colour + relight forward
//   CNR_NO_CHAIN_SDF         no_chain_sdf      per-layer launches instead of the chain-fused saving SDF forward
//   CNR_CHAIN_GRAD           chain_grad        opt-in: the chain-fused gradient chain (slower than its 8 launches; kept parity-green)
//   CNR_NO_FDW               no_fdw            separate layer and weight-gradient launches instead of layer_dw_kernel (and sweep0_dw_kernel)
//   CNR_FDW_SPLIT            fdw_split         the fused launches' partial-sum slots filled by the two separate kernels
//   CNR_NO_TOP_FUSE          no_top_fuse       the top SDF layer's backward as launches of its own
//   CNR_NO_HEAD_BWD          no_head_bwd       narrow GEMM launches instead of head_bwd_kernel (16-float cotangent rows out of the compositor)
//   CNR_NO_HEAD_FWD          no_head_fwd       the layer kernel for the 3-wide forward heads
//   CNR_NO_STRIP_BWD         no_strip_bwd      narrow layer launch + strip launch instead of strip_bwd_kernel
//   CNR_NO_SAMPLER_FUSE      no_sampler_fuse   merge / up_sample / embedding as three launches per up-sampling step
//   CNR_NO_SWEEP0            no_sweep0         separate launches instead of sweep0_dw_kernel
//   CNR_NO_NARROW_BWD        no_narrow_bwd     separate launches instead of the backward forms of narrow_bwd_kernel
//   CNR_NO_NARROW_DX         no_narrow_dx      the FP32-MFMA layer kernel for the product-only form (end of the forward gradient chain)
//   CNR_DISABLE_WS           disable_ws        the FP32-MFMA layer kernel instead of the weight-stationary split-f16 kernels
//   CNR_WS_GENERIC           ws_generic        the interpreted weight-stationary kernel for every view / epilogue combination
//   CNR_WS_NOSTREAM          ws_nostream       the general weight-stationary kernel instead of its stream form
//   CNR_WS_SERP=0            ws_serp           every layer launch walks its tiles upwards (default 1: alternating, +0.35 %)
//   CNR_DW_FP32              dw_fp32           FP32-MFMA weight-gradient kernel for the main tiles
//   CNR_DW_BF16              dw_bf16           split-bf16 weight-gradient kernel even when row scales are available
//   CNR_ROCTX                roctx             roctx ranges around the phases of the forward / backward calls
//   -- CNR_TUNING build only (constants otherwise) --
//   CNR_WS_KINDS (0xffff) CNR_WS_MINW (96) CNR_WS_WGS (256) CNR_WS_MINTPW (1) CNR_CHAIN_WGS (0 = default) CNR_CHAIN_SHAPE (0; 41 | 22 | 12)
//   CNR_CHAIN_FWD_RT (0; 4 | 2 | 1) CNR_CHAIN_SDF_MAXP (unbounded) CNR_FDW_DEEP (0; bit 0 / 1: deeper prefetch of the P / D waves)
//   CNR_FDW_NOREV (0) CNR_FDW_REVMODE (0) and the ABLATION words CNR_FDW_DBG / CNR_CHAIN_FWD_DBG (parts of a kernel switched off: WRONG results)
#pragma once

namespace cnr {

struct DebugFlags {
  bool no_fused = false, no_chain_fwd = false, no_chain_sdf = false, chain_grad = false, no_fdw = false, fdw_split = false, no_top_fuse = false,
       no_head_bwd = false, no_head_fwd = false, no_strip_bwd = false, no_sampler_fuse = false, no_sweep0 = false, no_narrow_bwd = false,
       no_narrow_dx = false, disable_ws = false, ws_generic = false, ws_nostream = false, dw_fp32 = false, dw_bf16 = false, roctx = false;
  int ws_serp = 1;
#ifdef CNR_TUNING
  int ws_kinds = 0xffff, ws_minw = 96, ws_wgs = 256, ws_mintpw = 1, chain_wgs = 0, chain_shape = 0, chain_fwd_rt = 0, fdw_deep = 0, fdw_norev = 0,
      fdw_revmode = 0, fdw_dbg = 0, chain_fwd_dbg = 0;
  long chain_sdf_maxp = 1L << 62;
#else
  static constexpr int ws_kinds = 0xffff, ws_minw = 96, ws_wgs = 256, ws_mintpw = 1, chain_wgs = 0, chain_shape = 0, chain_fwd_rt = 0, fdw_deep = 0,
                       fdw_norev = 0, fdw_revmode = 0, fdw_dbg = 0, chain_fwd_dbg = 0;
  static constexpr long chain_sdf_maxp = 1L << 62;
#endif
};

// parsed on first use (thread-safe function-local static); defined in cnr_plan.cpp -- the library's only getenv site
const DebugFlags& debug_flags();

// an ablation word inside a kernel: the kernel argument in the tuning build, the constant 0 (branches fold away) in the product build
#ifdef CNR_TUNING
#define CNR_ABLATION(x) (x)
#else
#define CNR_ABLATION(x) 0
#endif

}  // namespace cnr
