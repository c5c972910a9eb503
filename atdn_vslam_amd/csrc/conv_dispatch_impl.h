// Included only by the conv_inst_*.hip translation units.
#pragma once
#include "conv_dispatch.h"

namespace atdn {

template <class Epi>
inline void set_groups(Epi&, int) {}
inline void set_groups(EpiBiasStats& ep, int groups) { ep.groups_per_img = groups; }

template <int MODE, class Epi>
TileChoice conv_dispatch(const ConvShape& s, Epi ep, hipStream_t st) {
  const int Ho = conv_out(s.H, s.KH, s.stride, s.padH), Wo = conv_out(s.W, s.KW, s.stride, s.padW);
  TileChoice t = choose_tile(s.nimg, Ho * Wo, s.N);
  t.groups_per_img = cdiv(Ho * Wo, t.BM) * (t.BM / 32);
  set_groups(ep, t.groups_per_img);
  if (t.BM == 128 && t.BN == 128) launch_conv<MODE, 2, 2, 2, 2>(s, ep, st);
  else if (t.BM == 128 && t.BN == 64) launch_conv<MODE, 2, 1, 2, 2>(s, ep, st);
  else if (t.BM == 128 && t.BN == 96) launch_conv<MODE, 1, 3, 4, 1>(s, ep, st);
  else if (t.BM == 128 && t.BN == 32) launch_conv<MODE, 1, 1, 4, 1>(s, ep, st);
  else launch_conv<MODE, 1, 1, 2, 2>(s, ep, st);
  return t;
}

#define ATDN_INSTANTIATE_CONV(MODE, EPI) \
  template TileChoice conv_dispatch<MODE, EPI>(const ConvShape&, EPI, hipStream_t);

}  // namespace atdn
