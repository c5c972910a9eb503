// Implicit-GEMM convolution / NT-GEMM engine on the exact-fp32 matrix core
// (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[img][pix][n] = epilogue( sum_k A[img][pix][k] * Wt[n][k] )
//
// * Activations are NHWC. A rows are gathered on the fly (im2col never exists):
//   TAP mode  — channel counts are multiples of 32; a K-chunk of 32 floats is one
//               filter tap x 32 channels of ONE input pixel; up to two source
//               tensors form a "virtual concat" along channels (GRU hx, r*h‖x).
//   ROW mode  — thin-channel layers (C = 4 or 16, power of two, ld == C): for a
//               fixed filter row ky the taps (kx, c) are one contiguous run of
//               KW*C floats in NHWC memory, padded to a multiple of 32 (the pad
//               hits zero weights); a chunk's eight float4 slots may belong to
//               different pixels, so bounds are checked per slot.
// * Weights are [N][ldw] rows, K-contiguous, packed to the same K order. A
//   per-image weight stride makes the same kernel a batched NT-GEMM
//   (all-pairs correlation, Q·K^T, attention·V).
// * M tiles never straddle images (tile -> (img, pix0)), so spatial halos,
//   instance-norm statistics and per-pair operands need no special cases.
// * 256 threads = 4 waves in a WGM x WGN grid; each wave owns TM x TN MFMA tiles
//   of 32x32. K is permuted inside every 8-wide group (lane half h, step s ->
//   k = 4h + s) so each lane fetches its four A (or B) operands with a single
//   conflict-free ds_read_b128 from a [row][36]-float LDS image.
// * fp32 MFMA is exact-fp32 (bitwise an fmaf chain) at 64 cycles per 32x32x2:
//   the loop is matrix-pipe bound as long as the next chunk's global loads are
//   in flight during the current chunk's 16*TM*TN MFMAs — done here with a
//   register prefetch (global -> VGPR during compute, VGPR -> LDS after).
#pragma once
#include "common.h"

namespace atdn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_TAP = 0, MODE_ROW = 1 };

struct ConvGeom {
  const float* src0;
  const float* src1;
  long sb0, sb1;  // per-image strides (floats)
  int ld0, ld1;   // pixel strides (floats)
  int C0, C1;     // TAP: channels from each source (multiples of 32); ROW: C0 = channels per pixel
  int H, W, Ho, Wo;
  int KH, KW, stride, padH, padW;
  int cpk;      // K-chunks per filter row
  int nchunks;  // KH * cpk
  int log2C;    // ROW mode
  const float* w;
  long wb;  // per-image weight stride (0: shared)
  int ldw;
  int N;  // valid output channels == valid weight rows
  int tiles_per_img, nimg, ntile_n;
};

constexpr int LDS_LD = 36;  // 32 + 4 floats: ds_read_b128 rows land on distinct bank quads

template <int MODE, int TM, int TN, int WGM, int WGN, class Epi>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvGeom g, const Epi ep) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int RA = BM / 32, RB = BN / 32;  // float4 loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
  float* As = lds;
  float* Bs = lds + BM * LDS_LD;

  const int tid = threadIdx.x;
  const int nblk = g.nimg * g.tiles_per_img * g.ntile_n;
  const int id = xcd_remap(blockIdx.x, nblk);
  const int tile_n = id % g.ntile_n;
  const int tmg = id / g.ntile_n;
  const int img = tmg / g.tiles_per_img;
  const int pix0 = (tmg % g.tiles_per_img) * BM;
  const int n0 = tile_n * BN;
  const int HoWo = g.Ho * g.Wo;

  // ---- loader role: thread -> (row r0 (+32 i), float4 slot s)
  const int s = tid & 7;
  const int r0 = tid >> 3;
  int iy0[RA], ix0[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = pix0 + r0 + 32 * i;
    if (m < HoWo) {
      const int oy = m / g.Wo, ox = m - oy * g.Wo;
      iy0[i] = oy * g.stride - g.padH;
      ix0[i] = ox * g.stride - g.padW;
    } else {
      iy0[i] = -(1 << 20);  // every tap fails the bounds test
      ix0[i] = -(1 << 20);
    }
  }
  // Loads are unconditional (clamped address, value zeroed by a select afterwards): a predicated load makes hipcc
  // branch around it and wait vmcnt(0) per load, which serialises the whole prefetch.
  const float* wrow[RB];
  bool wok[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + r0 + 32 * j;
    wok[j] = n < g.N;
    wrow[j] = g.w + (long)img * g.wb + (long)(wok[j] ? n : 0) * g.ldw + 4 * s;
  }
  const float* s0 = g.src0 + (long)img * g.sb0;
  const float* s1 = (MODE == MODE_TAP && g.src1) ? g.src1 + (long)img * g.sb1 : nullptr;

  float4 ra[RA], rb[RB];
  bool aok[RA];  // validity of the prefetched A rows: applied when the registers are written to LDS, AFTER the
                 // MFMA block, so that nothing consumes a load result (and waits for it) before the matrix work
  int ky = 0, kx = 0, cc = 0;  // TAP: tap (ky,kx), channel offset cc; ROW: ky, part = kx
  const int ctot = g.C0 + g.C1;

  auto fetch = [&](int q) __attribute__((always_inline)) {
    if (MODE == MODE_TAP) {
      const float* sp;
      int ld, co;
      if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int iy = iy0[i] + ky, ix = ix0[i] + kx;
        const bool ok = ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
        const float4 v = *reinterpret_cast<const float4*>(sp + (long)(ok ? iy * g.W + ix : 0) * ld + co + 4 * s);
        ra[i] = v;
        aok[i] = ok;
      }
      cc += 32;
      if (cc == ctot) { cc = 0; if (++kx == g.KW) { kx = 0; ++ky; } }
    } else {
      const int f = kx * 32 + 4 * s;  // float index inside the (padded) row run
      const int dx = f >> g.log2C, c = f & (g.C0 - 1);
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int iy = iy0[i] + ky, ix = ix0[i] + dx;
        const bool ok = ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
        const float4 v = *reinterpret_cast<const float4*>(s0 + (long)(ok ? iy * g.W + ix : 0) * g.ld0 + c);
        ra[i] = v;
        aok[i] = ok;
      }
      if (++kx == g.cpk) { kx = 0; ++ky; }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(wrow[j] + q * 32);
      rb[j] = v;
    }
  };

  // ---- MFMA role
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const float* a_rd = As + (wm * TM * 32 + r) * LDS_LD + 4 * h;
  const float* b_rd = Bs + (wn * TN * 32 + r) * LDS_LD + 4 * h;

  fetch(0);
  for (int q = 0; q < g.nchunks; ++q) {
    __syncthreads();  // previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *reinterpret_cast<float4*>(As + (r0 + 32 * i) * LDS_LD + 4 * s) = keep_if(aok[i], ra[i]);
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<float4*>(Bs + (r0 + 32 * j) * LDS_LD + 4 * s) = keep_if(wok[j], rb[j]);
    __syncthreads();
    if (q + 1 < g.nchunks) fetch(q + 1);  // in flight during the MFMAs below
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(a_rd + i * 32 * LDS_LD + kc * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(b_rd + j * 32 * LDS_LD + kc * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: C/D map of the 32x32 tile: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  // per-column constants once per wave, before any store (a load issued after a store waits for that store too)
  typename EpiCol<Epi>::type colj[TN];
  float biasj[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = min(n0 + (wn * TN + j) * 32 + r, g.N - 1);
    biasj[j] = 0.f;
    if constexpr (Epi::kStats) biasj[j] = ep.bias[n];
    if constexpr (epi_bias_arg<Epi>::value) colj[j] = ep.col(n);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mbase = pix0 + (wm * TM + i) * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + r;
      const bool nok = n < g.N;
      if constexpr (Epi::kStats) {
        // per-column (sum, M2) of this 32-row group for instance norm, combined across the two lane halves
        const float bias = nok ? biasj[j] : 0.f;
        float v[16];
        float sum = 0.f;
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          v[e] = acc[i][j][e] + bias;
          if (m < HoWo) { sum += v[e]; ++cnt; }
        }
        sum += __shfl_xor(sum, 32);
        cnt += __shfl_xor(cnt, 32);
        const float mean = sum / (float)(cnt > 0 ? cnt : 1);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (m < HoWo) { const float d = v[e] - mean; m2 += d * d; }
        }
        m2 += __shfl_xor(m2, 32);
        if (h == 0 && nok) {
          const int grp = mbase >> 5;
          const long o = ((long)img * ep.groups_per_img + grp) * g.N + n;
          ep.part_sum[o] = sum;
          ep.part_m2[o] = m2;
        }
      }
      if (nok) {
        if constexpr (Epi::kPrefetch) {
          // all 16 operand loads of the tile are issued back to back (clamped rows), then applied
          typename Epi::Aux aux[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) aux[e] = ep.load(img, min(mbase + (e & 3) + 8 * (e >> 2) + 4 * h, HoWo - 1), n);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < HoWo) ep.apply(img, m, n, acc[i][j][e], aux[e]);
          }
        } else if constexpr (epi_bias_arg<Epi>::value) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < HoWo) ep.store_c(img, m, n, acc[i][j][e], colj[j]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < HoWo) ep(img, m, n, acc[i][j][e]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ host side
struct ConvShape {  // what the caller describes; ConvGeom is derived from it
  const float* src0 = nullptr; const float* src1 = nullptr;
  long sb0 = 0, sb1 = 0;
  int ld0 = 0, ld1 = 0, C0 = 0, C1 = 0;
  int H = 1, W = 1;
  int KH = 1, KW = 1, stride = 1, padH = 0, padW = 0;
  const float* w = nullptr; long wb = 0; int ldw = 0;
  const float* wfrag16 = nullptr;   // ... in the operand order of the 16x16x32 MFMA (pack_fragment_major16), optional
  int N = 0;
  int nimg = 1;
  // optional (generation-6 halo kernels with a statistics epilogue only): src0 is the RAW fp32 output of an
  // InstanceNorm'ed layer and these are its per-(image, channel) mean / reciprocal std: the loader normalises + ReLUs
  const float* in_mean = nullptr; const float* in_rstd = nullptr;
};

inline int conv_out(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

template <int MODE>
inline ConvGeom make_geom(const ConvShape& s, int BM, int BN) {
  ConvGeom g{};
  g.src0 = s.src0; g.src1 = s.src1; g.sb0 = s.sb0; g.sb1 = s.sb1; g.ld0 = s.ld0; g.ld1 = s.ld1;
  g.C0 = s.C0; g.C1 = s.C1; g.H = s.H; g.W = s.W;
  g.KH = s.KH; g.KW = s.KW; g.stride = s.stride; g.padH = s.padH; g.padW = s.padW;
  g.Ho = conv_out(s.H, s.KH, s.stride, s.padH);
  g.Wo = conv_out(s.W, s.KW, s.stride, s.padW);
  if (MODE == MODE_TAP) {
    ATDN_CHECK(s.C0 % 32 == 0 && s.C1 % 32 == 0 && s.C0 > 0, "TAP mode needs channel counts that are multiples of 32");
    ATDN_CHECK(s.ld0 % 4 == 0 && (s.C1 == 0 || (s.src1 && s.ld1 % 4 == 0)), "TAP mode needs 16-byte aligned pixels");
    g.cpk = s.KW * ((s.C0 + s.C1) / 32);
    g.log2C = 0;
  } else {
    ATDN_CHECK(s.C1 == 0 && s.src1 == nullptr, "ROW mode takes one source");
    ATDN_CHECK(s.C0 >= 4 && (s.C0 & (s.C0 - 1)) == 0 && s.ld0 == s.C0, "ROW mode needs dense power-of-two channels");
    g.cpk = cdiv(s.KW * s.C0, 32);
    int l = 0; while ((1 << l) < s.C0) ++l;
    g.log2C = l;
  }
  g.nchunks = s.KH * g.cpk;
  g.w = s.w; g.wb = s.wb; g.ldw = s.ldw; g.N = s.N;
  ATDN_CHECK(s.ldw % 4 == 0 && s.ldw >= g.nchunks * 32, "weight rows must hold the padded K and be 16-byte aligned");
  ATDN_CHECK(((uintptr_t)s.src0 % 16) == 0 && ((uintptr_t)s.w % 16) == 0 && s.sb0 % 4 == 0 && s.wb % 4 == 0,
             "operands must be 16-byte aligned");
  ATDN_CHECK(g.Ho > 0 && g.Wo > 0 && s.N > 0 && s.nimg > 0, "empty convolution");
  g.nimg = s.nimg;
  g.tiles_per_img = cdiv(g.Ho * g.Wo, BM);
  g.ntile_n = cdiv(s.N, BN);
  return g;
}

// K length (floats) of one packed weight row for a layer
inline int packed_k(int mode, int KH, int KW, int C) {
  return mode == MODE_TAP ? KH * KW * C : KH * round_up(KW * C, 32);
}

template <int MODE, int TM, int TN, int WGM, int WGN, class Epi>
inline void launch_conv(const ConvShape& s, const Epi& ep, hipStream_t st) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  ConvGeom g = make_geom<MODE>(s, BM, BN);
  const int nblk = g.nimg * g.tiles_per_img * g.ntile_n;
  hipLaunchKernelGGL((conv_mfma_kernel<MODE, TM, TN, WGM, WGN, Epi>), dim3(nblk), dim3(256), 0, st, g, ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
