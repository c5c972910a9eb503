#include "conv_sf_dispatch_impl.h"
#include "small_convs.h"
namespace atdn {
// flow head conv1 with conv2's partial sums fused into its epilogue (epilogues_sf.h: SfFlowHeadPartial): always the
// 128-wide block, two blocks = all output channels of their 128 pixels
void launch_flow_head_fused(const ConvShape& s, float wscale, const SfFlowHeadPartial& ep, hipStream_t st) {
  ATDN_CHECK(s.N == 256 && s.KH == 3 && s.KW == 3 && s.wfrag16 != nullptr, "flow head conv1 is 3x3 x 256 channels");
  // Round 5: two 128-channel blocks of four waves per pixel tile instead of one 256-channel block of eight. The eight-wave block
  // was alone on its CU (230 registers) with sixteen block barriers in its epilogue; two four-wave blocks share a CU like the
  // ConvGRU kernels' and each writes its own copy of the 18 partial sums, which flow_gather_kernel adds in channel order:
  // 2.35 / 2.37 -> 2.24 / 2.23 ms per forward (profiles/r05_ab_flow_head_128.txt).
  // (the f16 fast mode measures 4 % better on the old 256-wide block — 1.155 against 1.20 ms per forward; one shape for both modes)
  ATDN_CHECK(ep.gstride > 0, "two channel blocks per pixel tile write two copies of G");
  if (sf_fast_mode()) launch_conv_sf6_m<8, 128, 1, 4, 3, 3, SfFlowHeadPartial, true, false>(s, wscale, ep, st);
  else launch_conv_sf6_m<8, 128, 1, 4, 3, 3, SfFlowHeadPartial, false, false>(s, wscale, ep, st);
}
// (always the 128-wide block, whatever the batch: results must not depend on how many pairs share a launch — at one pair it
// covers the chip worse than the 64-wide blocks the dispatch would pick, a few us per iteration of the single-pair forward)
}  // namespace atdn
