#include "conv_sf_dispatch_impl.h"
#include "small_convs.h"
namespace atdn {
// flow head conv1 with conv2's partial sums fused into its epilogue (epilogues_sf.h: SfFlowHeadPartial): always the
// 256-wide block, one block = all output channels of its 128 pixels
void launch_flow_head_fused(const ConvShape& s, float wscale, const SfFlowHeadPartial& ep, hipStream_t st) {
  ATDN_CHECK(s.N == 256 && s.KH == 3 && s.KW == 3 && s.wfrag16 != nullptr, "flow head conv1 is 3x3 x 256 channels");
  if (sf_fast_mode()) launch_conv_sf6_m<8, 256, 1, 8, 3, 3, SfFlowHeadPartial, true, false>(s, wscale, ep, st);
  else launch_conv_sf6_m<8, 256, 1, 8, 3, 3, SfFlowHeadPartial, false, false>(s, wscale, ep, st);
}
// (always the 256-wide block, whatever the batch: results must not depend on how many pairs share a launch — at one pair it
// covers the chip worse than the 64-wide blocks the dispatch would pick, a few us per iteration of the single-pair forward)
}  // namespace atdn
