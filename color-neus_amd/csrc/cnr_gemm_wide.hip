// FP32-MFMA layer GEMM, 160..256-column tiles (only reached when the weight-stationary kernel is switched off or not applicable).
#include "cnr_gemm_fp32.h"

namespace cnr {

// One instantiation (8 column tiles) serves 5..8: the surplus tiles multiply weight rows past round_up(N, 32) (whatever follows in
// the arena, always mapped) into columns that the epilogue drops; this path is a debugging fallback, not a hot path.
void launch_layer_gemm_wide(const LayerGemm& g, int nt, cnr_stream s) {
  (void)nt;
  launch_layer_gemm<8>(g, s);
}

}  // namespace cnr
