// Weight-stationary layer GEMM, instantiations with store-type epilogues (forward SDF chain, gradient chain).
#include "cnr_gemm_ws.h"

namespace cnr {

bool ws_launch_group_a(const LayerGemm& g, int wrows, cnr_stream s) {
  const int vk = g.A.kind, ek = g.E.kind;
  const bool plain = g.E.tail_src == nullptr && g.E.split == (1 << 30);
#define WS_CASE(V_, E_)                                              \
  if (vk == V_ && ek == E_) {                                        \
    if (plain) launch_ws_t<V_, E_, true>(g, wrows, s);               \
    else launch_ws_t<V_, E_, false>(g, wrows, s);                    \
    return true;                                                     \
  }
  WS_CASE(VK_SOFTPLUS, EK_STORE) WS_CASE(VK_DIRECT, EK_STORE) WS_CASE(VK_SOFTPLUS, EK_SDF_TOP)
  WS_CASE(VK_SIGMUL, EK_STORE) WS_CASE(VK_SIGMUL_ROW, EK_STORE) WS_CASE(VK_SIGMUL, EK_SPLIT)
#undef WS_CASE
  return false;
}

}  // namespace cnr
