// Included only by the conv_sf_inst_*.hip translation units.
#pragma once
#include "conv_dispatch_impl.h"
#include "conv_sf.h"
#include "conv_sf2.h"
#include "conv_sf3.h"
#include "conv_sf4.h"
#include "conv_sf6.h"
#include "conv_sfd.h"
#include "epilogues_sf.h"

namespace atdn {

template <class Epi>
TileChoice conv_sf_dispatch(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
  const int Ho = conv_out(s.H, s.KH, s.stride, s.padH), Wo = conv_out(s.W, s.KW, s.stride, s.padW);
  static const bool use_v2 = !(getenv("ATDN_NO_HALO") && getenv("ATDN_NO_HALO")[0] == '1');
  if (use_v2 && conv_sf2_eligible(s)) {
    // halo-patch kernel. 16x16 output tiles x 128 channels (512 threads, half the weight traffic per FLOP) when
    // that grid still covers the chip, else 8x16 tiles with 128- or 64-wide N tiles.
    // generation 4 (weights by LDS-DMA) is the default; ATDN_CONV_GEN=2 (register-staged weights) and =3 (warp-
    // specialised) select the other kernels, all within a few % of each other (tools/microbench_conv.py)
    // generation 6 (fragment-major weights straight to registers, no per-step barrier) serves 3x3 / 1x5 / 5x1 kernels
    // of the epilogues that opt in (kGen6); everything else, and ATDN_CONV_GEN=4, stays on generation 4
    static const int gen = getenv("ATDN_CONV_GEN") ? atoi(getenv("ATDN_CONV_GEN")) : 6;
    if (gen >= 6) {
      int bn = 0, th = 8;
      if (conv_sf6_try(s, wscale, ep, st, &bn, &th, sf_fast_mode())) return TileChoice{th * 16, bn, cdiv(Wo, 16) * cdiv(Ho, th) * (th / 2), true};
    }
    ATDN_CHECK(s.in_mean == nullptr, "normalise-on-load is served by the generation-6 statistics kernels only");
    static const int big_min = getenv("ATDN_BIG_TILE_MIN") ? atoi(getenv("ATDN_BIG_TILE_MIN")) : 224;
    const int tiles16 = s.nimg * cdiv(Wo, 16) * cdiv(Ho, 16);
    if (s.N > 64 && (long)tiles16 * cdiv(s.N, 128) >= big_min) {
      TileChoice t3{256, 128, cdiv(Wo, 16) * cdiv(Ho, 16) * 8, true};
      if (gen == 2) launch_conv_sf2<2, Epi, 16>(s, wscale, ep, st);
      else launch_conv_sf4<2, Epi, 16>(s, wscale, ep, st);
      return t3;
    }
    const int tiles = s.nimg * cdiv(Wo, 16) * cdiv(Ho, 8);
    TileChoice t2{128, 64, cdiv(Wo, 16) * cdiv(Ho, 8) * 4, true};
    const bool wide = s.N > 64 && (long)tiles * cdiv(s.N, 128) >= 400;
    if (wide) t2.BN = 128;
    if (gen == 3)      { if (wide) launch_conv_sf3<2>(s, wscale, ep, st); else launch_conv_sf3<1>(s, wscale, ep, st); }
    else if (gen == 2) { if (wide) launch_conv_sf2<2>(s, wscale, ep, st); else launch_conv_sf2<1>(s, wscale, ep, st); }
    else               { if (wide) launch_conv_sf4<2>(s, wscale, ep, st); else launch_conv_sf4<1>(s, wscale, ep, st); }
    return t2;
  }
  ATDN_CHECK(s.in_mean == nullptr, "normalise-on-load is served by the generation-6 statistics kernels only");
  TileChoice t = choose_tile(s.nimg, Ho * Wo, s.N);
  t.groups_per_img = cdiv(Ho * Wo, t.BM) * (t.BM / 32);
  set_groups(ep, t.groups_per_img);
  // conv_sfd (both tiles by LDS-DMA) measured 2-4 % slower than the register-staged conv_sf on these GEMM / 1x1 /
  // strided shapes (the A tile streams from HBM or needs per-lane padding logic either way): opt-in only
  static const bool dma = getenv("ATDN_GEMM_DMA") && getenv("ATDN_GEMM_DMA")[0] == '1';
  if (dma) {
    if (t.BM == 128 && t.BN == 128) launch_conv_sfd<2, 2, 2, 2>(s, wscale, ep, st);
    else if (t.BM == 128 && t.BN == 64) launch_conv_sfd<2, 1, 2, 2>(s, wscale, ep, st);
    else if (t.BM == 128 && t.BN == 96) launch_conv_sfd<1, 3, 4, 1>(s, wscale, ep, st);
    else if (t.BM == 128 && t.BN == 32) launch_conv_sfd<1, 1, 4, 1>(s, wscale, ep, st);
    else launch_conv_sfd<1, 1, 2, 2>(s, wscale, ep, st);
    return t;
  }
  // attention x V: the A operand (the attention matrix) is streamed once from HBM: three chunks of loads in flight
  // (measured per forward of 8 pairs: depth 1 5.16 ms, 2 5.03, 3 5.01, 4 5.55 — the fourth register set costs occupancy)
  if constexpr (std::is_same_v<Epi, SfAggregate>) {
    static const int pf = getenv("ATDN_AGG_PREFETCH") ? atoi(getenv("ATDN_AGG_PREFETCH")) : 3;
    if (pf > 1 && t.BM == 128 && t.BN == 128 && s.KH == 1 && s.KW == 1 && (s.C0 + s.C1) >= 2048) {
      if (sf_fast_mode()) launch_conv_sf<2, 2, 2, 2, Epi, true, 3>(s, wscale, ep, st);
      else launch_conv_sf<2, 2, 2, 2, Epi, false, 3>(s, wscale, ep, st);
      return t;
    }
  }
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

#define ATDN_INSTANTIATE_CONV_SF(EPI) \
  template TileChoice conv_sf_dispatch<EPI>(const ConvShape&, float, EPI, hipStream_t);

}  // namespace atdn
