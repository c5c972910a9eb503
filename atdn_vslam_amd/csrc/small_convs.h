// Dedicated kernels for the two convolutions of the GRU iteration with a degenerate GEMM shape (small_convs.hip).
#pragma once
#include "common.h"
#include "epilogues_sf.h"

namespace atdn {

// motion encoder convf1: flow4 NHWC4 [nimg][H][W][4] (channels 0, 1) -> relu(conv7x7 + bias), sf [nimg][H*W][128].
// wfrag: split-f16 fragment-major weights [wave 0..3][channel block 0..1][step 0..3][hi, lo][lane] x 16 B (gma.hip:
// pack_convf1_sf), pre-multiplied by 1 / wscale.
void launch_flow_conv7_sf(const float* flow4, int nimg, int H, int W, const float* wfrag, float wscale, const float* bias,
                          float* out_sf, bool fast, hipStream_t st);
// Flow head with conv2 folded into conv1's epilogue (epilogues_sf.h: SfFlowHeadPartial; conv_sf_inst_e.hip):
// conv1 writes G[img][pix][tap * 2 + output] = conv2's weights x relu(conv1) of THAT pixel; the gather sums each
// pixel's 3 x 3 neighbourhood (zero padding outside the map) and applies SfFlowDelta (conv2's bias, coordinate update).
struct ConvShape;
void launch_flow_head_fused(const ConvShape& s, float wscale, const SfFlowHeadPartial& ep, hipStream_t st);
void launch_flow_gather(const float* G, long gstride, int nimg, int H, int W, const SfFlowDelta& ep, hipStream_t st);

}  // namespace atdn
