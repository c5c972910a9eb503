// Split-f16 implicit-GEMM / batched NT-GEMM with BOTH operand tiles delivered by LDS-DMA (generation "d").
//
// Same tiling, gather, epilogues and MFMA order as conv_sf.h; the A (activation) and W tiles of every 32-channel
// K-chunk are written into double-buffered, unpadded LDS images by `global_load_lds_dwordx4` (1 KiB = 8 rows x 128 B
// per wave instruction), one barrier per chunk. 16-byte slots are XOR-swizzled with (row >> 1) & 7 on the per-lane
// source address and on the fragment reads (conflict-free ds_read_b128 groups). Zero padding: a lane whose tap falls
// outside the image (or whose row is beyond the tile's valid rows) reads from a 128-byte line of zeros in global
// memory instead, so the DMA itself writes the padding; weight rows beyond N are clamped (never stored).
#pragma once
#include "conv_dispatch.h"
#include "sf.h"

namespace atdn {

const float* zero_line();  // 256 bytes of zeros in device memory (kernels.hip)

template <int TM, int TN, int WGM, int WGN, class Epi>
__global__ __launch_bounds__(256) void conv_sfd_kernel(const ConvGeom g, const float wscale, const float* zline,
                                                       const Epi ep) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int RA = BM / 32, RB = BN / 32;  // 8-row DMA pieces per wave per chunk (4 waves)
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * 32];
  float* As = lds;                 // [2][BM][32 floats]
  float* Bs = lds + 2 * BM * 32;   // [2][BN][32 floats]

  const int tid = threadIdx.x;
  const int nblk = g.nimg * g.tiles_per_img * g.ntile_n;
  const int id = xcd_remap(blockIdx.x, nblk);
  const int tile_n = id % g.ntile_n;
  const int tmg = id / g.ntile_n;
  const int img = tmg / g.tiles_per_img;
  const int pix0 = (tmg % g.tiles_per_img) * BM;
  const int n0 = tile_n * BN;
  const int HoWo = g.Ho * g.Wo;

  const int lane = tid & 63, wave = tid >> 6;
  // ---- DMA roles: piece i of this wave covers tile rows [rb, rb+8); lane -> row rb + (lane>>3), physical slot lane&7,
  // logical slot (lane&7) ^ ((row>>1)&7)
  int iy0[RA], ix0[RA], asl[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int row = (wave * RA + i) * 8 + (lane >> 3);
    const int m = pix0 + row;
    asl[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    if (m < HoWo) {
      const int oy = m / g.Wo, ox = m - oy * g.Wo;
      iy0[i] = oy * g.stride - g.padH;
      ix0[i] = ox * g.stride - g.padW;
    } else {
      iy0[i] = -(1 << 20);
      ix0[i] = -(1 << 20);
    }
  }
  const float* wsrc[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int row = (wave * RB + j) * 8 + (lane >> 3);
    const int n = min(n0 + row, g.N - 1);
    wsrc[j] = g.w + (long)img * g.wb + (long)n * g.ldw + ((lane & 7) ^ ((row >> 1) & 7)) * 4;
  }
  const float* s0 = g.src0 + (long)img * g.sb0;
  const float* s1 = g.src1 ? g.src1 + (long)img * g.sb1 : nullptr;
  int ky = 0, kx = 0, cc = 0;
  const int ctot = g.C0 + g.C1;

  auto dma = [&](int q, int buf) {
    const float* sp;
    int ld, co;
    if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
      const float* src = ok ? sp + (long)(iy * g.W + ix) * ld + co + asl[i] : zline + asl[i];
      float* dst = As + (buf * BM + (wave * RA + i) * 8) * 32;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    cc += 32;
    if (cc == ctot) { cc = 0; if (++kx == g.KW) { kx = 0; ++ky; } }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      float* dst = Bs + (buf * BN + (wave * RB + j) * 8) * 32;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + q * 32),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  const int wm = wave / WGN, wn = wave % WGN;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: row base (multiple of 32) + r, logical slot 2t+h (hi) / 4+2t+h (lo), physical = logical ^ ((r>>1)&7)
  const int sw = (r >> 1) & 7;
  const char* a_rd = reinterpret_cast<const char*>(As) + (wm * TM * 32 + r) * 128;
  const char* b_rd = reinterpret_cast<const char*>(Bs) + (wn * TN * 32 + r) * 128;

  dma(0, 0);
  __syncthreads();  // vmcnt(0) + barrier: chunk 0 has landed
  for (int q = 0; q < g.nchunks; ++q) {
    const int P = q & 1;
    if (q + 1 < g.nchunks) dma(q + 1, 1 - P);  // buffer 1-P was last read in chunk q-1 (barrier since)
    const char* ap = a_rd + P * BM * 128;
    const char* bp = b_rd + P * BN * 128;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
      const int oh = ((2 * t + h) ^ sw) << 4, ol = ((4 + 2 * t + h) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(ap + i * 32 * 128 + oh);
        al[i] = *reinterpret_cast<const f16x8*>(ap + i * 32 * 128 + ol);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(bp + j * 32 * 128 + oh);
        bl[j] = *reinterpret_cast<const f16x8*>(bp + j * 32 * 128 + ol);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (q + 1 < g.nchunks) __syncthreads();  // publishes chunk q+1 (DMA drained by the barrier's vmcnt(0))
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mbase = pix0 + (wm * TM + i) * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + r;
      const bool nok = n < g.N;
      if constexpr (Epi::kStats) {
        const float bias = nok ? ep.bias[n] : 0.f;
        float v[16];
        float sum = 0.f;
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
          v[e] = acc[i][j][e] * wscale + bias;
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
            if (m < HoWo) ep.apply(img, m, n, acc[i][j][e] * wscale, aux[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < HoWo) ep(img, m, n, acc[i][j][e] * wscale);
          }
        }
      }
    }
  }
}

template <int TM, int TN, int WGM, int WGN, class Epi>
inline void launch_conv_sfd(const ConvShape& s, float wscale, const Epi& ep, hipStream_t st) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  ConvGeom g = make_geom<MODE_TAP>(s, BM, BN);
  const int nblk = g.nimg * g.tiles_per_img * g.ntile_n;
  hipLaunchKernelGGL((conv_sfd_kernel<TM, TN, WGM, WGN, Epi>), dim3(nblk), dim3(256), 0, st, g, wscale, zero_line(), ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
