// Included only by the conv_sf_inst_*.hip translation units.
#pragma once
#include "conv_dispatch_impl.h"
#include "conv_sf.h"
#include "epilogues_sf.h"

namespace atdn {

template <class Epi>
TileChoice conv_sf_dispatch(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
  const int Ho = conv_out(s.H, s.KH, s.stride, s.padH), Wo = conv_out(s.W, s.KW, s.stride, s.padW);
  const TileChoice t = choose_tile(s.nimg, Ho * Wo, s.N);
  set_groups(ep, cdiv(Ho * Wo, t.BM) * (t.BM / 32));
  if (t.BM == 128 && t.BN == 128) launch_conv_sf<2, 2, 2, 2>(s, wscale, ep, st);
  else if (t.BM == 128 && t.BN == 64) launch_conv_sf<2, 1, 2, 2>(s, wscale, ep, st);
  else if (t.BM == 128 && t.BN == 96) launch_conv_sf<1, 3, 4, 1>(s, wscale, ep, st);
  else if (t.BM == 128 && t.BN == 32) launch_conv_sf<1, 1, 4, 1>(s, wscale, ep, st);
  else launch_conv_sf<1, 1, 2, 2>(s, wscale, ep, st);
  return t;
}

#define ATDN_INSTANTIATE_CONV_SF(EPI) \
  template TileChoice conv_sf_dispatch<EPI>(const ConvShape&, float, EPI, hipStream_t);

}  // namespace atdn
