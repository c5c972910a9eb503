// Two convolutions of the GRU iteration that the MFMA engines serve badly (profiles/r01_v3_kernel_stats_1stream.csv):
//   * motion encoder convf1 (update.py:88): 7x7, 2 -> 128 channels on the flow field. K = 98: the ROW-mode fp32 MFMA
//     kernel pads every kernel row to 32 and ran 51 us; here K is padded to 128 on the split-f16 engine, weights in registers.
//   * flow head conv2 (update.py:12): 3x3, 256 -> 2 channels. N = 2: an MFMA block pads N to 32 (65 us); here one
//     v_dot2_f32_f16 chain per (pixel, output) on the split-f16 operands (hi*hi + hi*lo + lo*hi, fp32 accumulate).
#include <algorithm>
#include "small_convs.h"
#include "sf.h"

namespace atdn {

// ---------------------------------------------------------------------------------------------- 7x7, 2 -> 128
// Round 4: on the split-f16 matrix engine (the register-tiled fp32 VALU kernel of rounds 1-3 took 55 us per 16 pairs: 6,272
// FMAs per thread, 196 LDS weight reads per thread and 50 KB of weights staged per block). K = 7 rows x (7 taps x 2 channels)
// is padded to 8 x (8 x 2) = 128 = four steps of v_mfma_f32_16x16x32_f16, one step = TWO filter rows:
//   * lane (n = lane & 15, g = lane >> 4) of the pixel operand holds, for output pixel (row, col0 + n), filter row
//     2 s + (g >> 1), taps 4 (g & 1) .. + 3, both channels = four adjacent input pixels of an f16x2 plane: 16 contiguous
//     bytes of LDS (hi plane, lo plane), 4-byte aligned. The eighth tap and the eighth row meet zero weights.
//   * weights are the ROW operand (a lane ends up with 4 consecutive channels of one pixel) and live in REGISTERS for the
//     whole kernel: a wave owns 32 output channels = 2 channel blocks x 4 steps x (hi, lo) = 16 fragments = 64 registers,
//     loaded once (fragment-major copy, gma.hip: pack_convf1_sf). Blocks are persistent over 2 x 16-pixel tiles: per tile a
//     wave reads 16 small pixel fragments from LDS and issues 48 MFMAs; nothing else moves.
//   * the 9 x 24-pixel input patch of the next tile is fetched before the current tile is multiplied and split into the
//     other pair of planes afterwards: one barrier per tile.
constexpr int FC_PR = 2 + 6 + 1;     // patch rows: 2 output rows + 6 halo + the padded eighth filter row
constexpr int FC_PP = 24;            // patch pixels per row: 16 + 7 taps (the padded eighth included) = 23 -> 24
template <bool FAST>
__global__ __launch_bounds__(256, 3) void flow_conv7_sf_kernel(const float4* __restrict__ flow4, int nimg, int H, int W,
                                                               const float* __restrict__ wfrag, float wscale,
                                                               const float* __restrict__ bias, float* __restrict__ out, long ob,
                                                               int tiles_x, int tiles_img) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ unsigned int plane[2][2][FC_PR * FC_PP];   // [buffer][hi | lo][pixel] = f16x2 (flow x, flow y)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, g16 = lane >> 4;
  const int ntiles = nimg * tiles_img;

  // this wave's weights: [channel block][step], hi and lo
  f16x8 wh[2][4], wl[2][4];
  {
    const char* wb = reinterpret_cast<const char*>(wfrag) + lane * 16;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const long f = (((long)wave * 2 + cb) * 4 + s) * 2;
        wh[cb][s] = *reinterpret_cast<const f16x8*>(wb + f * 1024);
        if (!FAST) wl[cb][s] = *reinterpret_cast<const f16x8*>(wb + (f + 1) * 1024);
      }
  }
  float4 b4[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) b4[cb] = *reinterpret_cast<const float4*>(bias + 32 * wave + 16 * cb + 4 * g16);

  // patch loader: thread -> one patch pixel
  const int ppy = tid / FC_PP, ppx = tid - ppy * FC_PP;
  const bool pin = tid < FC_PR * FC_PP;
  bool clamped = false;
  auto fetch = [&](int t) -> float2 {
    const int img = t / tiles_img, tloc = t - img * tiles_img;
    const int iy = (tloc / tiles_x) * 2 - 3 + ppy, ix = (tloc % tiles_x) * 16 - 3 + ppx;
    float2 v = make_float2(0.f, 0.f);
    if (pin && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      const float4 f = flow4[((long)img * H + iy) * W + ix];
      v = make_float2(f.x, f.y);
    }
    return v;
  };
  auto stash = [&](int buf, float2 v) {
    if (pin) {
      const SfPair x = sf_split_flag(v.x, clamped), y = sf_split_flag(v.y, clamped);
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      plane[buf][0][tid] = __builtin_bit_cast(unsigned int, f16x2{x.hi, y.hi});
      plane[buf][1][tid] = __builtin_bit_cast(unsigned int, f16x2{x.lo, y.lo});
    }
  };

  int t = blockIdx.x;
  if (t < ntiles) stash(0, fetch(t));
  __syncthreads();
  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    const int buf = it & 1;
    const int tn = t + gridDim.x;
    float2 nxt = make_float2(0.f, 0.f);
    if (tn < ntiles) nxt = fetch(tn);   // lands while this tile is multiplied

    f32x4v acc[2][2];   // [output row][channel block]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rb][cb][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int p = (rb + 2 * s + (g16 >> 1)) * FC_PP + n16 + 4 * (g16 & 1);
        u32x4 h4, l4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { h4[k] = plane[buf][0][p + k]; if (!FAST) l4[k] = plane[buf][1][p + k]; }
        const f16x8 ph = __builtin_bit_cast(f16x8, h4);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x4v c = acc[rb][cb];
          if (!FAST) {
            const f16x8 pl = __builtin_bit_cast(f16x8, l4);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cb][s], pl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cb][s], ph, c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cb][s], ph, c, 0, 0, 0);
          acc[rb][cb] = c;
        }
      }
    // lane (n, g) holds, for pixel (row rb, column n), channels 32 wave + 16 cb + 4 g + 0..3
    {
      const int img = t / tiles_img, tloc = t - img * tiles_img;
      const int oy0 = (tloc / tiles_x) * 2, ox = (tloc % tiles_x) * 16 + n16;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int oy = oy0 + rb;
        if (oy < H && ox < W) {
          const long off = (long)img * ob + ((long)oy * W + ox) * 128;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const float4 o = make_float4(fmaxf(acc[rb][cb][0] * wscale + b4[cb].x, 0.f), fmaxf(acc[rb][cb][1] * wscale + b4[cb].y, 0.f),
                                         fmaxf(acc[rb][cb][2] * wscale + b4[cb].z, 0.f), fmaxf(acc[rb][cb][3] * wscale + b4[cb].w, 0.f));
            sf_store4_flag(out, off, 32 * wave + 16 * cb + 4 * g16, o, clamped);
          }
        }
      }
    }
    if (tn < ntiles) stash(buf ^ 1, nxt);
    __syncthreads();   // the other pair of planes is published; this pair was last read before the barrier
  }
  sf_report(clamped);
}

void launch_flow_conv7_sf(const float* flow4, int nimg, int H, int W, const float* wfrag, float wscale, const float* bias,
                          float* out_sf, bool fast, hipStream_t st) {
  const int tx = cdiv(W, 16), ty = cdiv(H, 2);
  const int ntiles = nimg * tx * ty;
  // (three persistent blocks per CU; four or eight per CU — the kernel's 126 registers allow four resident — measure the same
  // or 0.3 % worse on the motion-encoder stage, round 5)
  const int grid = std::min(ntiles, 256 * 3);
  if (fast)
    hipLaunchKernelGGL(flow_conv7_sf_kernel<true>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(flow4), nimg, H, W,
                       wfrag, wscale, bias, out_sf, (long)H * W * 128, tx, tx * ty);
  else
    hipLaunchKernelGGL(flow_conv7_sf_kernel<false>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(flow4), nimg, H, W,
                       wfrag, wscale, bias, out_sf, (long)H * W * 128, tx, tx * ty);
  ATDN_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------- conv2 gather
// out[p][o] = sum over taps t = (ty, tx) of G[p + (ty - 1, tx - 1)][t * 2 + o]   (conv2 3x3, pad 1: pixels outside
// the map contribute nothing), then the flow update. One thread per pixel; G (4 MB at 8 pairs) is L2-resident.
__global__ __launch_bounds__(256) void flow_gather_kernel(const float* __restrict__ G, long gstride, int H, int W, long total,
                                                          const SfFlowDelta ep) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long HW = (long)H * W;
  const int img = (int)(i / HW), m = (int)(i - img * HW);
  const int y = m / W, x = m - y * W;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int sy = y + t / 3 - 1, sx = x + t % 3 - 1;
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
      const float* gp = G + ((long)img * HW + (long)sy * W + sx) * 18 + 2 * t;
      float2 g = *reinterpret_cast<const float2*>(gp);
      if (gstride) {   // two channel blocks per pixel tile: their partial sums, in channel order
        const float2 g1 = *reinterpret_cast<const float2*>(gp + gstride);
        g.x += g1.x; g.y += g1.y;
      }
      s0 += g.x; s1 += g.y;
    }
  }
  const SfFlowDelta::Aux a0 = ep.load(img, m, 0), a1 = ep.load(img, m, 1);
  ep.apply(img, m, 0, s0, a0);
  ep.apply(img, m, 1, s1, a1);
}

void launch_flow_gather(const float* G, long gstride, int nimg, int H, int W, const SfFlowDelta& ep, hipStream_t st) {
  const long total = (long)nimg * H * W;
  hipLaunchKernelGGL(flow_gather_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, G, gstride, H, W, total, ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
