// Included only by the conv_sf_inst_*.hip translation units.
#pragma once
#include "conv_dispatch_impl.h"
#include "conv_sf.h"
#include "conv_sf6.h"
#include "epilogues_sf.h"

namespace atdn {

template <class Epi>
TileChoice conv_sf_dispatch(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
  const int Ho = conv_out(s.H, s.KH, s.stride, s.padH), Wo = conv_out(s.W, s.KW, s.stride, s.padW);
  // Stride-1 3x3 / 1x5 / 5x1 convolutions run on the halo-patch kernel (conv_sf6.h) for every epilogue that opts in
  // (kGen6); shapes it does not serve (and 1x1 / strided convolutions, the batched GEMMs of the correlation levels and the
  // projections) go to the plain implicit GEMM below.
  if (conv_halo_eligible(s)) {
    int bn = 0, th = 8;
    if (conv_sf6_try(s, wscale, ep, st, &bn, &th, sf_fast_mode())) return TileChoice{th * 16, bn, cdiv(Wo, 16) * cdiv(Ho, th) * (th / 2), true};
  }
  ATDN_CHECK(s.in_mean == nullptr, "normalise-on-load is served by the halo-patch statistics kernels only");
  TileChoice t = choose_tile(s.nimg, Ho * Wo, s.N);
  t.groups_per_img = cdiv(Ho * Wo, t.BM) * (t.BM / 32);
  set_groups(ep, t.groups_per_img);
  if (sf_fast_mode()) {
    if (t.BM == 128 && t.BN == 128) launch_conv_sf<2, 2, 2, 2, Epi, true>(s, wscale, ep, st);
    else if (t.BM == 128 && t.BN == 64) launch_conv_sf<2, 1, 2, 2, Epi, true>(s, wscale, ep, st);
    else if (t.BM == 128 && t.BN == 96) launch_conv_sf<1, 3, 4, 1, Epi, true>(s, wscale, ep, st);
    else if (t.BM == 128 && t.BN == 32) launch_conv_sf<1, 1, 4, 1, Epi, true>(s, wscale, ep, st);
    else launch_conv_sf<1, 1, 2, 2, Epi, true>(s, wscale, ep, st);
    return t;
  }
  if (t.BM == 128 && t.BN == 128) launch_conv_sf<2, 2, 2, 2>(s, wscale, ep, st);
  else if (t.BM == 128 && t.BN == 64) launch_conv_sf<2, 1, 2, 2>(s, wscale, ep, st);
  else if (t.BM == 128 && t.BN == 96) launch_conv_sf<1, 3, 4, 1>(s, wscale, ep, st);
  else if (t.BM == 128 && t.BN == 32) launch_conv_sf<1, 1, 4, 1>(s, wscale, ep, st);
  else launch_conv_sf<1, 1, 2, 2>(s, wscale, ep, st);
  return t;
}

// Two independent convolutions with the same epilogue class: one launch when the halo-patch path would give both the same kernel
// (conv_sf6_try_pair), two launches otherwise. Same results either way.
template <class Epi>
void conv_sf_dispatch_pair(const ConvShape& s0, float wscale0, Epi ep0, const ConvShape& s1, float wscale1, Epi ep1, hipStream_t st) {
  if (conv_sf6_try_pair(s0, wscale0, ep0, s1, wscale1, ep1, st, sf_fast_mode())) return;
  conv_sf_dispatch(s0, wscale0, ep0, st);
  conv_sf_dispatch(s1, wscale1, ep1, st);
}

#define ATDN_INSTANTIATE_CONV_SF(EPI) \
  template TileChoice conv_sf_dispatch<EPI>(const ConvShape&, float, EPI, hipStream_t);
#define ATDN_INSTANTIATE_CONV_SF_PAIR(EPI) \
  template void conv_sf_dispatch_pair<EPI>(const ConvShape&, float, EPI, const ConvShape&, float, EPI, hipStream_t);

}  // namespace atdn
