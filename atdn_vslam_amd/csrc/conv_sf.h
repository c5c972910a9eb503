// Implicit-GEMM convolution / batched NT-GEMM on the f16 matrix core with split-f16 ("sf", sf.h) operands:
//   A·B ≈ A_hi·B_hi + A_hi·B_lo + A_lo·B_hi        (three v_mfma_f32_16x16x32_f16, fp32 accumulate)
// f16 x f16 products are exact in fp32 and the dropped lo·lo term is 2^-22 relative, so the result is fp32-grade
// while the matrix pipe runs at 16/3 = 5.3x the rate of v_mfma_f32_32x32x2_f32.
//
// Same tiling, gather (TAP mode only: channel counts are multiples of 32), virtual concat, per-image M tiling,
// XCD remap and epilogue interface as conv_mfma.h. Differences:
// * operands are sf tensors: a 128-byte K-chunk of a pixel is [32 hi | 32 lo] halves, copied verbatim to LDS
//   ([row][160 B]); lane (n, g) reads the 16-byte slot g of row n's hi and lo halves with two ds_read_b128;
// * packed weights carry a per-layer power-of-two scale (max|w'| in [1,2)) so that the lo halves of small
//   weights stay normal f16 numbers; the accumulator is multiplied by `wscale` = 2^-p before the epilogue.
#pragma once
#include "conv_dispatch.h"
#include "sf.h"

namespace atdn {

// The products run on v_mfma_f32_16x16x32_f16 (K = 32 = one chunk per instruction; the chip holds a higher clock under this
// shape than under 32x32x16): lane (n = lane & 15, g = lane >> 4) holds row / column n of a 16-wide block and
// the 16-byte slot g of the chunk's [32 hi] / [32 lo] halves; the LDS row pitch is 160 B (conflict-free for that lane map);
// a 32 x 32 tile is 2 x 2 blocks of four accumulator registers: lane (n, g) holds column 16 cb + n and rows 16 hb + 4 g + 0..3.
template <int TM, int TN, int WGM, int WGN, class Epi, bool FAST = false>
__global__ __launch_bounds__(256) void conv_sf_kernel(const ConvGeom g, const float wscale, const Epi ep) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int LDS_LD = 40;   // floats per LDS row (160 B)
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
      iy0[i] = -(1 << 20);
      ix0[i] = -(1 << 20);
    }
  }
  // unconditional loads + select (see conv_mfma.h)
  const float* wrow[RB];
  bool wok[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + r0 + 32 * j;
    wok[j] = n < g.N;
    wrow[j] = g.w + (long)img * g.wb + (long)(wok[j] ? n : 0) * g.ldw + 4 * s;
  }
  const float* s0 = g.src0 + (long)img * g.sb0;
  const float* s1 = g.src1 ? g.src1 + (long)img * g.sb1 : nullptr;

  constexpr int PF = 1;   // one chunk of global loads in flight ahead of the one being multiplied
  float4 ra[PF][RA], rb[PF][RB];
  bool aok[PF][RA];  // applied at LDS-store time (see conv_mfma.h)
  int ky = 0, kx = 0, cc = 0;
  const int ctot = g.C0 + g.C1;

  auto fetch = [&](int q, int slot) __attribute__((always_inline)) {
    const float* sp;
    int ld, co;
    if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
      const float* ap = sp + (long)(ok ? iy * g.W + ix : 0) * ld + co + 4 * s;
      ra[slot][i] = *reinterpret_cast<const float4*>(ap);
      aok[slot][i] = ok;
    }
    cc += 32;
    if (cc == ctot) { cc = 0; if (++kx == g.KW) { kx = 0; ++ky; } }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(wrow[j] + q * 32);
      rb[slot][j] = v;
    }
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  // accumulators: b[row block][column block] x 4
  struct Acc16 { f32x4v b[2][2]; };
  Acc16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j].b[e >> 3][(e >> 2) & 1][e & 3] = 0.f;
    }

  // byte view of the LDS image: hi halves at [0,64), lo halves at [64,128) of a row
  constexpr int ROWB = LDS_LD * 4;
  const int rr = lane & 15, hh = lane >> 4;
  const char* a_rd = reinterpret_cast<const char*>(As + (wm * TM * 32 + rr) * LDS_LD) + 16 * hh;
  const char* b_rd = reinterpret_cast<const char*>(Bs + (wn * TN * 32 + rr) * LDS_LD) + 16 * hh;

  // The chunk loop is branch-free: look-ahead fetches past the end are clamped (valid, unused data) and the trip
  // count is rounded up to a multiple of PF with the surplus chunks stored to LDS as zeros, so the waits the compiler
  // places before a slot's LDS stores are exact counts of the younger loads, never a drain.
  const int last = g.nchunks - 1;
#pragma unroll
  for (int d = 0; d < PF; ++d) fetch(min(d, last), d);
  const int nq = (g.nchunks + PF - 1) / PF * PF;
  for (int q0 = 0; q0 < nq; q0 += PF) {
#pragma unroll
   for (int d = 0; d < PF; ++d) {
    const int q = q0 + d;
    const bool live = PF == 1 || q <= last;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *reinterpret_cast<float4*>(As + (r0 + 32 * i) * LDS_LD + 4 * s) = keep_if(aok[d][i] && live, ra[d][i]);
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<float4*>(Bs + (r0 + 32 * j) * LDS_LD + 4 * s) = keep_if(wok[j], rb[d][j]);
    __syncthreads();
    fetch(min(q + PF, last), d);   // into the slot just drained
    f16x8 ah[TM][2], al[TM][2], bh[TN][2], bl[TN][2];   // [tile][16-row block]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        ah[i][hb] = *reinterpret_cast<const f16x8*>(a_rd + (i * 32 + 16 * hb) * ROWB);
        if constexpr (!FAST) al[i][hb] = *reinterpret_cast<const f16x8*>(a_rd + (i * 32 + 16 * hb) * ROWB + 64);
      }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        bh[j][cb] = *reinterpret_cast<const f16x8*>(b_rd + (j * 32 + 16 * cb) * ROWB);
        if constexpr (!FAST) bl[j][cb] = *reinterpret_cast<const f16x8*>(b_rd + (j * 32 + 16 * cb) * ROWB + 64);
      }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x4v c = acc[i][j].b[hb][cb];
            if constexpr (!FAST) {
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i][hb], bh[j][cb], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i][hb], bl[j][cb], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i][hb], bh[j][cb], c, 0, 0, 0);
            acc[i][j].b[hb][cb] = c;
          }
   }
  }

  // Epilogue: a lane owns NSET = 2 channel columns of a 32 x 32 tile with NPX = 8 rows each: columns 16 cs + (lane & 15), rows
  // 16 (e >> 2) + 4 (lane >> 4) + (e & 3).
  constexpr int NSET = 2, NPX = 8;
  auto col_of = [&](int cs) { return 16 * cs + (lane & 15); };
  auto row_of = [&](int e) { return 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3); };
  auto val_of = [&](int i, int j, int cs, int e) __attribute__((always_inline)) { return acc[i][j].b[e >> 2][cs][e & 3]; };
  // per-column constants once per wave, before any store (a load issued after a store waits for that store too)
  typename EpiCol<Epi>::type colj[TN][NSET];
  float biasj[TN][NSET];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int cs = 0; cs < NSET; ++cs) {
      const int n = min(n0 + (wn * TN + j) * 32 + col_of(cs), g.N - 1);
      biasj[j][cs] = 0.f;
      if constexpr (Epi::kStats) biasj[j][cs] = ep.bias[n];
      if constexpr (epi_bias_arg<Epi>::value) colj[j][cs] = ep.col(n);
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mbase = pix0 + (wm * TM + i) * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int cs = 0; cs < NSET; ++cs) {
        const int n = n0 + (wn * TN + j) * 32 + col_of(cs);
        const bool nok = n < g.N;
        if constexpr (Epi::kStats) {
          const float bias = nok ? biasj[j][cs] : 0.f;
          float v[NPX];
          float sum = 0.f;
          int cnt = 0;
#pragma unroll
          for (int e = 0; e < NPX; ++e) {
            const int m = mbase + row_of(e);
            v[e] = val_of(i, j, cs, e) * wscale + bias;
            if (m < HoWo) { sum += v[e]; ++cnt; }
          }
          sum += __shfl_xor(sum, 16);
          cnt += __shfl_xor(cnt, 16);
          sum += __shfl_xor(sum, 32);
          cnt += __shfl_xor(cnt, 32);
          const float mean = sum / (float)(cnt > 0 ? cnt : 1);
          float m2 = 0.f;
#pragma unroll
          for (int e = 0; e < NPX; ++e) {
            const int m = mbase + row_of(e);
            if (m < HoWo) { const float d = v[e] - mean; m2 += d * d; }
          }
          m2 += __shfl_xor(m2, 16);
          m2 += __shfl_xor(m2, 32);
          if ((lane >> 4) == 0 && nok) {
            const int grp = mbase >> 5;
            const long o = ((long)img * ep.groups_per_img + grp) * g.N + n;
            ep.part_sum[o] = sum;
            ep.part_m2[o] = m2;
          }
        }
        if (nok) {
          if constexpr (Epi::kPrefetch) {
            // all operand loads of the column set are issued back to back (clamped rows), then applied
            typename Epi::Aux aux[NPX];
#pragma unroll
            for (int e = 0; e < NPX; ++e) aux[e] = ep.load(img, min(mbase + row_of(e), HoWo - 1), n);
#pragma unroll
            for (int e = 0; e < NPX; ++e) {
              const int m = mbase + row_of(e);
              if (m < HoWo) ep.apply(img, m, n, val_of(i, j, cs, e) * wscale, aux[e]);
            }
          } else if constexpr (epi_bias_arg<Epi>::value) {
#pragma unroll
            for (int e = 0; e < NPX; ++e) {
              const int m = mbase + row_of(e);
              if (m < HoWo) ep.store_c(img, m, n, val_of(i, j, cs, e) * wscale, colj[j][cs]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < NPX; ++e) {
              const int m = mbase + row_of(e);
              if (m < HoWo) ep(img, m, n, val_of(i, j, cs, e) * wscale);
            }
          }
        }
      }
  }
}

template <int TM, int TN, int WGM, int WGN, class Epi, bool FAST = false>
inline void launch_conv_sf(const ConvShape& s, float wscale, const Epi& ep, hipStream_t st) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  ConvGeom g = make_geom<MODE_TAP>(s, BM, BN);
  const int nblk = g.nimg * g.tiles_per_img * g.ntile_n;
  hipLaunchKernelGGL((conv_sf_kernel<TM, TN, WGM, WGN, Epi, FAST>), dim3(nblk), dim3(256), 0, st, g, wscale, ep);
  ATDN_HIP(hipGetLastError());
}

// Plain-f16 arithmetic (precision mode 2) for the calling thread's sf convolutions: conv_sf_dispatch issues only the
// hi x hi MFMA of every product while this is set (GmaNet sets it around its forward).
bool& sf_fast_mode();

// Definitions are explicitly instantiated in conv_sf_inst_*.hip
template <class Epi>
TileChoice conv_sf_dispatch(const ConvShape& s, float wscale, Epi ep, hipStream_t st);
// two independent convolutions with the same epilogue class as one launch where the halo-patch path serves both with the same
// kernel (conv_sf6.h: conv_sf6_pair_kernel), as two launches otherwise; instantiated for SfBias<ACT_RELU>
template <class Epi>
void conv_sf_dispatch_pair(const ConvShape& s0, float wscale0, Epi ep0, const ConvShape& s1, float wscale1, Epi ep1, hipStream_t st);

}  // namespace atdn
