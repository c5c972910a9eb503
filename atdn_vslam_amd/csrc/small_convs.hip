// Two convolutions of the GRU iteration that the MFMA engines serve badly (profiles/r01_v3_kernel_stats_1stream.csv):
//   * motion encoder convf1 (update.py:88): 7x7, 2 -> 128 channels on the flow field. K = 98: the ROW-mode fp32 MFMA
//     kernel pads every kernel row to 32 and ran 51 us; a register-tiled fp32 VALU kernel does it in its FMA time.
//   * flow head conv2 (update.py:12): 3x3, 256 -> 2 channels. N = 2: an MFMA block pads N to 32 (65 us); here one
//     v_dot2_f32_f16 chain per (pixel, output) on the split-f16 operands (hi*hi + hi*lo + lo*hi, fp32 accumulate).
#include "small_convs.h"
#include "sf.h"

namespace atdn {

// ---------------------------------------------------------------------------------------------- 7x7, 2 -> 128
// block: 8x16 output pixels x 128 channels; thread: 8 pixels (half a tile row) x 8 channels. Every weight vector read
// from LDS feeds 8 pixels (4 in the first version: its weight reads took as long as its FMAs)
__global__ __launch_bounds__(256) void flow_conv7_kernel(const float4* __restrict__ flow4, int H, int W,
                                                         const float* __restrict__ wl /*[98][128]*/,
                                                         const float* __restrict__ bias, float* __restrict__ out, long ob,
                                                         int tiles_x, int tiles_img) {
  __shared__ __attribute__((aligned(16))) float ws[98 * 128];
  __shared__ float2 ps[14][33];   // pitch 66 dwords = 2 (mod 64 banks): the 8 rows x 2 column groups one ds_read_b64 touches are 16 distinct bank pairs (pitch 24: rows r and r + 4 collide)
  const int tid = threadIdx.x;
  const int img = blockIdx.x / tiles_img, tloc = blockIdx.x - img * tiles_img;
  const int ty0 = (tloc / tiles_x) * 8, tx0 = (tloc % tiles_x) * 16;
  for (int i = tid; i < 98 * 128 / 4; i += 256)
    reinterpret_cast<float4*>(ws)[i] = reinterpret_cast<const float4*>(wl)[i];
  for (int i = tid; i < 14 * 22; i += 256) {
    const int py = i / 22, px = i - py * 22;
    const int iy = ty0 - 3 + py, ix = tx0 - 3 + px;
    float2 v = make_float2(0.f, 0.f);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      const float4 f = flow4[((long)img * H + iy) * W + ix];
      v = make_float2(f.x, f.y);
    }
    ps[py][px] = v;
  }
  __syncthreads();
  const int pg = tid & 15, cg = tid >> 4;
  const int row = pg >> 1, xg = pg & 1;
  const int ch0 = cg * 8;
  float acc[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    float2 in[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) in[i] = ps[row + ky][xg * 8 + i];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4* wp = reinterpret_cast<const float4*>(ws + ((ky * 7 + kx) * 2 + c) * 128 + ch0);
        const float4 w0 = wp[0], w1 = wp[1];
        const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = c ? in[j + kx].y : in[j + kx].x;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][e] += v * w8[e];
        }
      }
  }
  const float4 b0 = *reinterpret_cast<const float4*>(bias + ch0), b1 = *reinterpret_cast<const float4*>(bias + ch0 + 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int oy = ty0 + row, ox = tx0 + xg * 8 + j;
    if (oy < H && ox < W) {
      const long off = (long)img * ob + ((long)oy * W + ox) * 128;
      sf_store4(out, off, ch0, make_float4(fmaxf(acc[j][0] + b0.x, 0.f), fmaxf(acc[j][1] + b0.y, 0.f),
                                           fmaxf(acc[j][2] + b0.z, 0.f), fmaxf(acc[j][3] + b0.w, 0.f)));
      sf_store4(out, off, ch0 + 4, make_float4(fmaxf(acc[j][4] + b1.x, 0.f), fmaxf(acc[j][5] + b1.y, 0.f),
                                               fmaxf(acc[j][6] + b1.z, 0.f), fmaxf(acc[j][7] + b1.w, 0.f)));
    }
  }
}

void launch_flow_conv7(const float* flow4, int nimg, int H, int W, const float* wl, const float* bias, float* out_sf,
                       hipStream_t st) {
  const int tx = cdiv(W, 16), ty = cdiv(H, 8);
  hipLaunchKernelGGL(flow_conv7_kernel, dim3(nimg * tx * ty), dim3(256), 0, st, reinterpret_cast<const float4*>(flow4), H, W,
                     wl, bias, out_sf, (long)H * W * 128, tx, tx * ty);
  ATDN_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------- conv2 gather
// out[p][o] = sum over taps t = (ty, tx) of G[p + (ty - 1, tx - 1)][t * 2 + o]   (conv2 3x3, pad 1: pixels outside
// the map contribute nothing), then the flow update. One thread per pixel; G (4 MB at 8 pairs) is L2-resident.
__global__ __launch_bounds__(256) void flow_gather_kernel(const float* __restrict__ G, int H, int W, long total,
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
      const float2 g = *reinterpret_cast<const float2*>(G + ((long)img * HW + (long)sy * W + sx) * 18 + 2 * t);
      s0 += g.x; s1 += g.y;
    }
  }
  const SfFlowDelta::Aux a0 = ep.load(img, m, 0), a1 = ep.load(img, m, 1);
  ep.apply(img, m, 0, s0, a0);
  ep.apply(img, m, 1, s1, a1);
}

void launch_flow_gather(const float* G, int nimg, int H, int W, const SfFlowDelta& ep, hipStream_t st) {
  const long total = (long)nimg * H * W;
  hipLaunchKernelGGL(flow_gather_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, G, H, W, total, ep);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
