// Split-f16 implicit-GEMM convolution, generation 2: spatially blocked with an LDS-resident halo patch.
//
// conv_sf.h re-reads every A row once per filter tap (9x for a 3x3), which makes the f16 matrix pipe wait on
// L2. Here an M tile is a TH x TW block of OUTPUT pixels of one image (TH*TW = 128); per 32-channel chunk the
// (TH+KH-1) x (TW+KW-1) input patch (true zero padding included) is staged in LDS ONCE and all KH*KW taps are
// served from it by offsetting the fragment address — A traffic drops by ~KH*KW/1.4. Weight tiles [BN][32-chunk]
// stream per (chunk, tap) through a double-buffered LDS image with a register prefetch: one barrier per tap,
// two at a chunk boundary. Stride-1 TAP-mode convolutions only (everything at 1/8 resolution and the stride-1
// encoder convs); 1x1 / strided / GEMM shapes stay on conv_sf_kernel.
#pragma once
#include "conv_sf.h"

namespace atdn {

struct Conv2Geom {
  const float* src0; const float* src1;
  long sb0, sb1;
  int ld0, ld1, C0, C1;
  int H, W, Ho, Wo;
  int KH, KW, padH, padW;
  int PH, PW;            // patch size in pixels
  int tiles_x, tiles_y;  // per image
  int nimg, ntile_n;
  const float* w; int ldw; int N;
  float wscale;
  // normalise-on-load (conv_sf6.h NORM): src0 is RAW fp32 [pix][C0] and the patch loader applies
  // relu((x - in_mean[img][c]) * in_rstd[img][c]) before splitting to sf (InstanceNorm + ReLU of the producer)
  const float* in_mean; const float* in_rstd;
};

// patch pixels an TH x TW output tile needs for the supported filters (3x3, 1x5, 5x1): 8x16 -> 192, 16x16 -> 324
constexpr int c2_patch_max(int TH, int TW) {
  int a = (TH + 2) * (TW + 2), b = TH * (TW + 4), c = (TH + 4) * TW;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
constexpr int C2_PATCH_MAX = c2_patch_max(8, 16);

// An M tile is TH x TW output pixels handled by TH*TW/32 row-tiles: TH*TW/64 x 2 waves (2 x TN MFMA tiles each),
// NT = TH*TW*2 threads. 8x16 (256 threads, 2 blocks per CU) for small grids; 16x16 (512 threads, 1 block per CU)
// halves the weight-tile traffic per FLOP — with 128x128 tiles the loop needs the whole L2 gather bandwidth
// (27 B/clk/CU at full matrix rate), which is what bounds it.
template <int TH, int TW, int TN, class Epi>
__global__ __launch_bounds__(TH * TW * 2) void conv_sf2_kernel(const Conv2Geom g, const Epi ep) {
  static_assert((TH * TW) % 128 == 0, "M tile is a multiple of 128 output pixels");
  constexpr int NT = TH * TW * 2;
  constexpr int RSTEP = NT / 8;                  // LDS rows covered by one pass of the loader threads
  constexpr int PMAX = c2_patch_max(TH, TW);
  constexpr int BN = 64 * TN;
  constexpr int RB = BN / RSTEP;                 // weight float4 per thread per step
  static_assert(RB >= 1 && BN % RSTEP == 0, "weight tile must be a whole number of loader passes");
  constexpr int NP = (PMAX + RSTEP - 1) / RSTEP; // patch float4 per thread per chunk
  constexpr int ROWB = LDS_LD * 4;
  __shared__ __attribute__((aligned(16))) float lds[(PMAX + 2 * BN) * LDS_LD];
  float* Ps = lds;
  float* Ws = lds + PMAX * LDS_LD;

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

  // ---- loader roles
  const int s = tid & 7, r0 = tid >> 3;
  int poff[NP];  // input pixel index of this thread's patch rows, -1 = zero padding / unused
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
  // unconditional loads + select at LDS-store time (see conv_mfma.h); weight row addresses are recomputed per
  // fetch from one base pointer (clamped to the last valid row) instead of living in 2*RB registers
  const int nrow0 = n0 + r0;
  const float* wbase = g.w + 4 * s;
  const float* s0 = g.src0 + (long)img * g.sb0;
  const float* s1 = g.src1 ? g.src1 + (long)img * g.sb1 : nullptr;
  const int nck = (g.C0 + g.C1) >> 5, ntap = g.KH * g.KW;
  const int nstep = nck * ntap;

  float4 pr[NP], wrA[RB], wrB[RB];  // weight tiles are prefetched TWO steps ahead, alternating register sets
  auto fetch_patch = [&](int c) {
    const int cc = c << 5;
    const float* sp; int ld, co;
    if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(sp + (long)(poff[k] >= 0 ? poff[k] : 0) * ld + co + 4 * s);
      pr[k] = v;
    }
  };
  // linear step index st = c * ntap + tap; the packed K order is [tap][channel chunk]
  auto fetch_w = [&](float4 (&wr)[RB], int st) {
    const int c = st / ntap, tap = st - c * ntap;
    const int q = tap * nck + c;
#pragma unroll
    for (int j = 0; j < RB; ++j)
      wr[j] = *reinterpret_cast<const float4*>(wbase + (long)min(nrow0 + RSTEP * j, g.N - 1) * g.ldw + q * 32);
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (r0 + RSTEP * k < PMAX)
        *reinterpret_cast<float4*>(Ps + (r0 + RSTEP * k) * LDS_LD + 4 * s) = keep_if(poff[k] >= 0, pr[k]);
  };
  auto store_w = [&](const float4 (&wr)[RB], int buf) {
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<float4*>(Ws + (buf * BN + r0 + RSTEP * j) * LDS_LD + 4 * s) = keep_if(nrow0 + RSTEP * j < g.N, wr[j]);  // the select also gives the
      // LDS store its own source registers: storing straight from wr[] makes the refill of wr[] (issued right
      // after) wait for the store to drain (measured: -28 % whole-forward throughput)
  };

  // ---- MFMA roles: wave grid 2 x 2, each wave 2 x TN tiles of 32x32
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
  int a_off[2];  // byte offset of this lane's A row inside the patch for tap (0,0)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (wm * 2 + i) * 32 + r;
    a_off[i] = ((p / TW) * g.PW + (p % TW)) * ROWB + 16 * h;
  }
  const char* Pb = reinterpret_cast<const char*>(Ps);
  const char* Wb = reinterpret_cast<const char*>(Ws) + (wn * TN * 32 + r) * ROWB + 16 * h;

  fetch_patch(0);
  fetch_w(wrA, 0);
  if (nstep > 1) fetch_w(wrB, 1);
  store_patch();
  store_w(wrA, 0);
  __syncthreads();
  if (nstep > 2) fetch_w(wrA, 2);
  int tap = 0, c = 0, ky = 0, kx = 0;
  // One step = one (channel chunk, filter tap). Step st (parity P) computes from weight buffer P, then stores
  // the register set that holds W(st+1) — issued two steps earlier — into buffer 1-P and refills it with W(st+3).
  // (Plain loop with a runtime parity: a generic lambda kept its by-reference captures in scratch memory, and
  // every scratch reload is a VMEM op whose wait drains the whole prefetch queue.)
  for (int st = 0; st < nstep; ++st) {
    const int P = st & 1;  // wave-uniform: selects the weight buffer and which register set is stored / refilled
    if (tap == 0 && c + 1 < nck) fetch_patch(c + 1);  // lands during this chunk's taps
    const char* arow = Pb + (ky * g.PW + kx) * ROWB;
    const char* brow = Wb + P * BN * ROWB;
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
        bh[j] = *reinterpret_cast<const f16x8*>(brow + j * 32 * ROWB + 32 * t);
        bl[j] = *reinterpret_cast<const f16x8*>(brow + j * 32 * ROWB + 32 * t + 64);
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
      if (P == 0) { store_w(wrB, 1); if (st + 3 < nstep) fetch_w(wrB, st + 3); }
      else        { store_w(wrA, 0); if (st + 3 < nstep) fetch_w(wrA, st + 3); }
      __syncthreads();
    }
    if (last_tap) { tap = 0; ++c; } else ++tap;
    if (++kx == g.KW) { kx = 0; if (++ky == g.KH) ky = 0; }
  }

  // ---- epilogue
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
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (mm[e] >= 0) ep(img, mm[e], n, acc[i][j][e] * g.wscale);
        }
      }
    }
  }
}

// true if the shape can run on conv_sf2_kernel
inline bool conv_sf2_eligible(const ConvShape& s) {
  if (s.stride != 1 || s.KH * s.KW == 1 || s.wb != 0) return false;
  const int PH = 8 + s.KH - 1, PW = 16 + s.KW - 1;
  const int PH2 = 16 + s.KH - 1, PW2 = 16 + s.KW - 1;
  return PH * PW <= c2_patch_max(8, 16) && PH2 * PW2 <= c2_patch_max(16, 16);
}

template <int TN, class Epi, int TH = 8>
inline void launch_conv_sf2(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
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
  hipLaunchKernelGGL((conv_sf2_kernel<TH, TW, TN, Epi>), dim3(nblk), dim3(TH * TW * 2), 0, st, g, ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
