// Split-f16 halo-patch convolution, generation 4: weight tiles arrive by LDS-DMA.
//
// Same tiling, patch staging, MFMA order and epilogues as conv_sf2.h. The difference is the weight path: instead of
// global_load -> VGPR -> select -> ds_write_b128 (ablation on the z|r ConvGRU conv: +18 us for the LDS stores and
// +45 us for the loads on a 115 us MFMA stream) each wave issues `global_load_lds_dwordx4` pieces (1 KiB = 8 weight
// rows x 128 B per instruction) straight into a double-buffered LDS image. A DMA writes lane-linear bytes, so rows are
// unpadded (128 B) and bank conflicts are avoided by an XOR swizzle of the 16-byte slot with (row >> 1) & 7, applied
// to the per-lane SOURCE address and to the fragment reads (the 16 lanes of every ds_read_b128 group then hit 16
// distinct slots). Rows beyond N are clamped reads of valid memory: their accumulators are never stored, so the
// weight path needs no zeroing at all.
#pragma once
#include "conv_sf2.h"

namespace atdn {

template <int TH, int TW, int TN, class Epi>
__global__ __launch_bounds__(TH * TW * 2) void conv_sf4_kernel(const Conv2Geom g, const Epi ep) {
  static_assert((TH * TW) % 128 == 0, "M tile is a multiple of 128 output pixels");
  constexpr int NT = TH * TW * 2;
  constexpr int NW = NT / 64;
  constexpr int RSTEP = NT / 8;
  constexpr int PMAX = c2_patch_max(TH, TW);
  constexpr int BN = 64 * TN;
  constexpr int NI = BN / 8 / NW;  // DMA instructions per wave per weight tile
  static_assert(NI >= 1 && BN % (8 * NW) == 0, "weight tile must split into whole 8-row DMA pieces per wave");
  constexpr int NP = (PMAX + RSTEP - 1) / RSTEP;
  constexpr int ROWB = LDS_LD * 4;
  __shared__ __attribute__((aligned(16))) float lds[PMAX * LDS_LD + 2 * BN * 32];
  float* Ps = lds;
  float* Ws = lds + PMAX * LDS_LD;  // [2][BN][32 floats]

  const int tid = threadIdx.x;
  const int tiles_img = g.tiles_x * g.tiles_y;
  const int nblk = g.nimg * tiles_img * g.ntile_n;
  const int id = xcd_remap(blockIdx.x, nblk);
  const int tile_n = id % g.ntile_n;
  const int tmg = id / g.ntile_n;
  const int img = tmg / tiles_img;
  const int tloc = tmg - img * tiles_img;
  const int ty0 = (tloc / g.tiles_x) * TH, tx0 = (tloc % g.tiles_x) * TW;
  const int n0 = tile_n * BN;
  const int npatch = g.PH * g.PW;
  const int lane = tid & 63, wave = tid >> 6;

  // ---- patch loader role (registers, true zero padding)
  const int s = tid & 7, r0 = tid >> 3;
  int poff[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int prow = r0 + RSTEP * k;
    int off = -1;
    if (prow < npatch) {
      const int py = prow / g.PW, px = prow - py * g.PW;
      const int iy = ty0 - g.padH + py, ix = tx0 - g.padW + px;
      if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) off = iy * g.W + ix;
    }
    poff[k] = off;
  }
  const float* s0 = g.src0 + (long)img * g.sb0;
  const float* s1 = g.src1 ? g.src1 + (long)img * g.sb1 : nullptr;
  const int nck = (g.C0 + g.C1) >> 5, ntap = g.KH * g.KW;
  const int nstep = nck * ntap;

  // ---- weight DMA role: piece i of this wave covers tile rows [rb, rb + 8); lane -> (row rb + (lane>>3), physical
  // slot lane&7) and fetches the logical slot (lane&7) ^ ((row>>1)&7) of that row
  const float* wsrc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rb = (wave * NI + i) * 8;
    const int row = rb + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    wsrc[i] = g.w + (long)min(n0 + row, g.N - 1) * g.ldw + logical * 4;
  }
  auto dma_w = [&](int st, int buf) {
    const int c = st / ntap, tap = st - c * ntap;
    const int q = tap * nck + c;  // packed K order is [tap][channel chunk]
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float* dst = Ws + (buf * BN + (wave * NI + i) * 8) * 32;  // wave-uniform; the hardware adds lane * 16 bytes
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + q * 32),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  float4 pr[NP];
  auto fetch_patch = [&](int c) {
    const int cc = c << 5;
    const float* sp; int ld, co;
    if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
    for (int k = 0; k < NP; ++k)
      pr[k] = *reinterpret_cast<const float4*>(sp + (long)(poff[k] >= 0 ? poff[k] : 0) * ld + co + 4 * s);
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (r0 + RSTEP * k < PMAX)
        *reinterpret_cast<float4*>(Ps + (r0 + RSTEP * k) * LDS_LD + 4 * s) = keep_if(poff[k] >= 0, pr[k]);
  };

  // ---- MFMA roles
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (wm * 2 + i) * 32 + r;
    a_off[i] = ((p / TW) * g.PW + (p % TW)) * ROWB + 16 * h;
  }
  const char* Pb = reinterpret_cast<const char*>(Ps);
  // weight fragment: row (wn*TN + j)*32 + r, logical slot 2t + h (hi) / 4 + 2t + h (lo), physical = logical ^ ((r>>1)&7)
  const int wsw = (r >> 1) & 7;
  const char* Wb = reinterpret_cast<const char*>(Ws) + (wn * TN * 32 + r) * 128;

  fetch_patch(0);
  dma_w(0, 0);
  store_patch();
  __syncthreads();  // (waits for the DMA: vmcnt(0) + barrier)
  int tap = 0, c = 0, ky = 0, kx = 0;
  for (int st = 0; st < nstep; ++st) {
    const int P = st & 1;
    if (st + 1 < nstep) dma_w(st + 1, 1 - P);              // buffer 1-P was last read in step st-1 (barrier since)
    if (tap == 0 && c + 1 < nck) fetch_patch(c + 1);        // lands during this chunk's taps
    const char* arow = Pb + (ky * g.PW + kx) * ROWB;
    const char* brow = Wb + P * BN * 128;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f16x8 ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(arow + a_off[i] + 32 * t);
        al[i] = *reinterpret_cast<const f16x8*>(arow + a_off[i] + 32 * t + 64);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(brow + j * 32 * 128 + (((2 * t + h) ^ wsw) << 4));
        bl[j] = *reinterpret_cast<const f16x8*>(brow + j * 32 * 128 + (((4 + 2 * t + h) ^ wsw) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    const bool last_tap = (tap + 1 == ntap);
    if (st + 1 < nstep) {
      if (last_tap) {  // chunk boundary: the patch is replaced, every wave must be done reading it
        __syncthreads();
        store_patch();
      }
      __syncthreads();  // publishes W(st+1) (DMA drained by the barrier's vmcnt(0)) and the new patch
    }
    if (last_tap) { tap = 0; ++c; } else ++tap;
    if (++kx == g.KW) { kx = 0; if (++ky == g.KH) ky = 0; }
  }

  // ---- epilogue (as conv_sf2.h)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pbase = (wm * 2 + i) * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + r;
      const bool nok = n < g.N;
      int mm[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int p = pbase + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int oy = ty0 + p / TW, ox = tx0 + p % TW;
        mm[e] = (oy < g.Ho && ox < g.Wo) ? oy * g.Wo + ox : -1;
      }
      if constexpr (Epi::kStats) {
        const float bias = nok ? ep.bias[n] : 0.f;
        float v[16];
        float sum = 0.f;
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = acc[i][j][e] * g.wscale + bias;
          if (mm[e] >= 0) { sum += v[e]; ++cnt; }
        }
        sum += __shfl_xor(sum, 32);
        cnt += __shfl_xor(cnt, 32);
        const float mean = sum / (float)(cnt > 0 ? cnt : 1);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (mm[e] >= 0) { const float d = v[e] - mean; m2 += d * d; }
        m2 += __shfl_xor(m2, 32);
        const int grp = tloc * (TH * TW / 32) + wm * 2 + i;
        if (h == 0 && nok) {
          const long o = ((long)img * ep.groups_per_img + grp) * g.N + n;
          ep.part_sum[o] = sum;
          ep.part_m2[o] = m2;
        }
        if (lane == 0 && n == 0) ep.part_cnt[(long)img * ep.groups_per_img + grp] = (float)cnt;
      }
      if (nok) {
        if constexpr (Epi::kPrefetch) {
          typename Epi::Aux aux[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) aux[e] = ep.load(img, max(mm[e], 0), n);
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (mm[e] >= 0) ep.apply(img, mm[e], n, acc[i][j][e] * g.wscale, aux[e]);
        } else if constexpr (epi_bias_arg<Epi>::value) {
          const typename Epi::Col cn = ep.col(n);
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (mm[e] >= 0) ep.store_c(img, mm[e], n, acc[i][j][e] * g.wscale, cn);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (mm[e] >= 0) ep(img, mm[e], n, acc[i][j][e] * g.wscale);
        }
      }
    }
  }
}

template <int TN, class Epi, int TH = 8>
inline void launch_conv_sf4(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
  constexpr int TW = 16;
  Conv2Geom g{};
  g.src0 = s.src0; g.src1 = s.src1; g.sb0 = s.sb0; g.sb1 = s.sb1; g.ld0 = s.ld0; g.ld1 = s.ld1;
  g.C0 = s.C0; g.C1 = s.C1; g.H = s.H; g.W = s.W;
  g.KH = s.KH; g.KW = s.KW; g.padH = s.padH; g.padW = s.padW;
  g.Ho = conv_out(s.H, s.KH, 1, s.padH); g.Wo = conv_out(s.W, s.KW, 1, s.padW);
  g.PH = TH + s.KH - 1; g.PW = TW + s.KW - 1;
  ATDN_CHECK(conv_sf2_eligible(s), "shape not eligible for the halo-patch kernel");
  ATDN_CHECK(s.C0 % 32 == 0 && s.C1 % 32 == 0 && s.C0 > 0 && s.ld0 % 4 == 0, "TAP-mode channel constraints");
  ATDN_CHECK(s.ldw % 4 == 0 && s.ldw >= s.KH * s.KW * (s.C0 + s.C1), "weight rows too short");
  g.tiles_x = cdiv(g.Wo, TW); g.tiles_y = cdiv(g.Ho, TH);
  g.nimg = s.nimg; g.ntile_n = cdiv(s.N, 64 * TN);
  g.w = s.w; g.ldw = s.ldw; g.N = s.N; g.wscale = wscale;
  set_groups(ep, g.tiles_x * g.tiles_y * (TH * TW / 32));
  const int nblk = g.nimg * g.tiles_x * g.tiles_y * g.ntile_n;
  hipLaunchKernelGGL((conv_sf4_kernel<TH, TW, TN, Epi>), dim3(nblk), dim3(TH * TW * 2), 0, st, g, ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
