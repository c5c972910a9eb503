// Split-f16 halo-patch convolution, warp-specialised (generation 3).
//
// Same math, tiling and LDS images as conv_sf2.h, but a workgroup has 8 waves with fixed roles:
//   waves 0-3  CONSUMERS  — ds_read fragments + 3xf16 MFMA + epilogue; they issue no global load in the loop
//   waves 4-7  PRODUCERS  — global loads two steps ahead, zero padding, LDS stores, all address arithmetic
// One producer and one consumer wave share each SIMD, so the producer's VALU/VMEM/LDS-store issue fills the
// slots the consumer leaves while its MFMAs run on the matrix pipe (rocprofv3 on the single-role kernel: 24 %
// of wave cycles issuing ~100 VALU per 24 MFMA, 43 % parked on s_waitcnt/s_barrier).
// Synchronisation is a raw s_barrier after `s_waitcnt lgkmcnt(0)` only, so the producers' in-flight global
// loads (vmcnt) survive the barrier; both roles execute the same barrier sequence: one per (chunk, tap) step,
// two where the patch is replaced.
#pragma once
#include "conv_sf2.h"

namespace atdn {

#define ATDN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ABL (diagnostic builds only, results are wrong): bit0 skip the producers' global loads inside the loop, bit1 skip
// their LDS stores, bit2 consumers skip the fragment ds_reads, bit3 no barriers inside the loop.
template <int TH, int TW, int TN, class Epi, int ABL = 0>
__global__ __launch_bounds__(512, 4) void conv_sf3_kernel(const Conv2Geom g, const Epi ep) {
  static_assert(TH * TW == 128, "M tile is 128 output pixels");
  constexpr int BN = 64 * TN;
  constexpr int RB = BN / 32;
  constexpr int NP = (C2_PATCH_MAX * 8 + 255) / 256;
  constexpr int ROWB = LDS_LD * 4;
  __shared__ __attribute__((aligned(16))) float lds[(C2_PATCH_MAX + 2 * BN) * LDS_LD];
  float* Ps = lds;
  float* Ws = lds + C2_PATCH_MAX * LDS_LD;

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
  const int nck = (g.C0 + g.C1) >> 5, ntap = g.KH * g.KW;
  const int nstep = nck * ntap;

  if (tid >= 256) {
    // =========================================================== PRODUCER
    const int pt = tid - 256;
    const int s = pt & 7, r0 = pt >> 3;
    const int npatch = g.PH * g.PW;
    int poff[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int prow = r0 + 32 * k;
      int off = -1;
      if (prow < npatch) {
        const int py = prow / g.PW, px = prow - py * g.PW;
        const int iy = ty0 - g.padH + py, ix = tx0 - g.padW + px;
        if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) off = iy * g.W + ix;
      }
      poff[k] = off;
    }
    const int nrow0 = n0 + r0;
    const float* wbase = g.w + 4 * s;
    const float* s0 = g.src0 + (long)img * g.sb0;
    const float* s1 = g.src1 ? g.src1 + (long)img * g.sb1 : nullptr;
    float4 pr[NP], wrA[RB], wrB[RB];
    auto fetch_patch = [&](int c) {
      const int cc = c << 5;
      const float* sp; int ld, co;
      if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
      for (int k = 0; k < NP; ++k)
        pr[k] = *reinterpret_cast<const float4*>(sp + (long)(poff[k] >= 0 ? poff[k] : 0) * ld + co + 4 * s);
    };
    auto fetch_w = [&](float4 (&wr)[RB], int st) {
      const int c = st / ntap, tap = st - c * ntap;
      const int q = tap * nck + c;
#pragma unroll
      for (int j = 0; j < RB; ++j)
        wr[j] = *reinterpret_cast<const float4*>(wbase + (long)min(nrow0 + 32 * j, g.N - 1) * g.ldw + q * 32);
    };
    auto store_patch = [&]() {
#pragma unroll
      for (int k = 0; k < NP; ++k)
        *reinterpret_cast<float4*>(Ps + (r0 + 32 * k) * LDS_LD + 4 * s) = keep_if(poff[k] >= 0, pr[k]);
    };
    auto store_w = [&](const float4 (&wr)[RB], int buf) {
#pragma unroll
      for (int j = 0; j < RB; ++j)
        *reinterpret_cast<float4*>(Ws + (buf * BN + r0 + 32 * j) * LDS_LD + 4 * s) =
            keep_if(nrow0 + 32 * j < g.N, wr[j]);  // the select also decouples the store from the refilled registers
    };
    fetch_patch(0);
    fetch_w(wrA, 0);
    if (nstep > 1) fetch_w(wrB, 1);
    store_patch();
    store_w(wrA, 0);
    if (nstep > 2) fetch_w(wrA, 2);
    ATDN_LDS_BARRIER();  // step 0 operands visible
    int tap = 0, c = 0;
    for (int st = 0; st < nstep; ++st) {
      const int P = st & 1;
      if (!(ABL & 1) && tap == 0 && c + 1 < nck) fetch_patch(c + 1);
      if (st + 1 < nstep) {  // W(st+1) -> buffer 1-P (last read in step st-1), then refill that register set
        if (P == 0) { if (!(ABL & 2)) store_w(wrB, 1); if (!(ABL & 1) && st + 3 < nstep) fetch_w(wrB, st + 3); }
        else        { if (!(ABL & 2)) store_w(wrA, 0); if (!(ABL & 1) && st + 3 < nstep) fetch_w(wrA, st + 3); }
      }
      if (!(ABL & 8)) ATDN_LDS_BARRIER();  // end of step st: consumers are done with buffer P and, at the last tap, with the patch
      const bool last_tap = (tap + 1 == ntap);
      if (last_tap && st + 1 < nstep) {
        if (!(ABL & 2)) store_patch();
        if (!(ABL & 8)) ATDN_LDS_BARRIER();
      }
      if (last_tap) { tap = 0; ++c; } else ++tap;
    }
    return;
  }

  // ============================================================= CONSUMER
  const int lane = tid & 63, wave = tid >> 6;
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
  const char* Wb = reinterpret_cast<const char*>(Ws) + (wn * TN * 32 + r) * ROWB + 16 * h;

  ATDN_LDS_BARRIER();
  {
    int tap = 0, ky = 0, kx = 0;
    for (int st = 0; st < nstep; ++st) {
      const char* arow = Pb + (ky * g.PW + kx) * ROWB;
      const char* brow = Wb + (st & 1) * BN * ROWB;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f16x8 ah[2], al[2], bh[TN], bl[TN];
        if (ABL & 4) {  // diagnostic: fragments from registers (kept opaque so the MFMAs are not folded away)
#pragma unroll
          for (int i = 0; i < 2; ++i) { asm volatile("" : "=v"(ah[i])); asm volatile("" : "=v"(al[i])); }
#pragma unroll
          for (int j = 0; j < TN; ++j) { asm volatile("" : "=v"(bh[j])); asm volatile("" : "=v"(bl[j])); }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(arow + a_off[i] + 32 * t);
            al[i] = *reinterpret_cast<const f16x8*>(arow + a_off[i] + 32 * t + 64);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(brow + j * 32 * ROWB + 32 * t);
            bl[j] = *reinterpret_cast<const f16x8*>(brow + j * 32 * ROWB + 32 * t + 64);
          }
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
      if (!(ABL & 8)) ATDN_LDS_BARRIER();
      const bool last_tap = (tap + 1 == ntap);
      if (!(ABL & 8) && last_tap && st + 1 < nstep) ATDN_LDS_BARRIER();
      if (last_tap) tap = 0; else ++tap;
      if (++kx == g.KW) { kx = 0; if (++ky == g.KH) ky = 0; }
    }
  }

  // ---- epilogue (consumer waves only)
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
        const int grp = tloc * 4 + wm * 2 + i;
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
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (mm[e] >= 0) ep(img, mm[e], n, acc[i][j][e] * g.wscale);
        }
      }
    }
  }
}

template <int TN, class Epi, int ABL = 0>
inline void launch_conv_sf3(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
  constexpr int TH = 8, TW = 16;
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
  set_groups(ep, g.tiles_x * g.tiles_y * 4);
  const int nblk = g.nimg * g.tiles_x * g.tiles_y * g.ntile_n;
  hipLaunchKernelGGL((conv_sf3_kernel<TH, TW, TN, Epi, ABL>), dim3(nblk), dim3(512), 0, st, g, ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
