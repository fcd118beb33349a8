// Internal interface between the GEMM translation units (split so that the many template instantiations build in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include "cnr_backend.h"

namespace cnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ws_absmax4(const f4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// cnr_gemm_ws_b.hip: weight-stationary layer GEMM (dispatch over the instantiations of both ws translation units)
void launch_layer_gemm_ws(const LayerGemm& g, int wrows, cnr_stream s);
bool ws_k17_supported(const LayerGemm& g);
// cnr_gemm_wide.hip: FP32-MFMA layer GEMM with 5..8 column tiles
void launch_layer_gemm_wide(const LayerGemm& g, int nt, cnr_stream s);
// cnr_gemm_ws_a.hip: the store-type epilogues; returns false when the combination is not one of its instantiations
bool ws_launch_group_a(const LayerGemm& g, int wrows, cnr_stream s);

}  // namespace cnr
