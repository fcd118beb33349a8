// Weight-stationary layer GEMM, instantiations of the backward-type epilogues, the 17-block variant, the interpreted
// fallback, and the dispatch over all instantiations.
#include "cnr_gemm_ws.h"

namespace cnr {

// K in (256, 272]: only the combinations the plan needs are instantiated with the 17th k-block
bool ws_k17_supported(const LayerGemm& g) {
  const bool plain = g.E.tail_src == nullptr && g.E.split == (1 << 30);
  return plain && g.A.kind == VK_DIRECT && (g.E.kind == EK_RELU || g.E.kind == EK_VBACK);
}

void launch_layer_gemm_ws(const LayerGemm& g_in, int wrows, cnr_stream s) {
  // a launch whose 256 columns all lie below the split point never touches o2: it is a plain launch (the columns at and beyond the split
  // belong to the narrow launch that follows) and can take the specialised / stream instantiations
  LayerGemm g = g_in;
  if (g.E.tail_src == nullptr && g.E.split != (1 << 30) && g.col0 + 256 <= g.E.split &&
      (g.E.kind == EK_SPLIT || g.E.kind == EK_SDF_TOP || g.E.kind == EK_RELU_MASK || g.E.kind == EK_VBACK)) {
    g.E.split = 1 << 30;
    g.E.o2 = nullptr;
  }
  const bool generic_only = debug_flags().ws_generic;   // debugging aid: interpreted kernel for every combination
  const int vk = g.A.kind, ek = g.E.kind;
  const bool plain = g.E.tail_src == nullptr && g.E.split == (1 << 30);
  if (g.K > 256) {
    if (ek == EK_RELU) launch_ws_t<VK_DIRECT, EK_RELU, true, true>(g, wrows, s);
    else launch_ws_t<VK_DIRECT, EK_VBACK, true, true>(g, wrows, s);
    return;
  }
  if (!generic_only && ws_launch_group_a(g, wrows, s)) return;
#define WS_CASE(V_, E_)                                              \
  if (!generic_only && vk == V_ && ek == E_) {                       \
    if (plain) launch_ws_t<V_, E_, true>(g, wrows, s);               \
    else launch_ws_t<V_, E_, false>(g, wrows, s);                    \
    return;                                                          \
  }
  // the combinations the render plan issues on 256-wide layers (cnr_plan.cpp)
  WS_CASE(VK_DIRECT, EK_RELU) WS_CASE(VK_DIRECT, EK_RELU_MASK) WS_CASE(VK_DIRECT, EK_SPLIT)
  WS_CASE(VK_DIRECT, EK_SWEEP) WS_CASE(VK_DIRECT, EK_VBACK)
#undef WS_CASE
  launch_ws_t<-1, -1, false>(g, wrows, s);
}

}  // namespace cnr
