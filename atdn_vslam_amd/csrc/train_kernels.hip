// See train_kernels.h. Everything here is HBM- or latency-bound bookkeeping around the convolutions; kernels are
// written for coalesced 16-byte accesses (NHWC16 maps: one float4 per thread) and deterministic reductions
// (per-block partials + a second pass, no floating-point atomics).
#include "train_kernels.h"
#include "common.h"
#include <algorithm>
#include <cmath>

namespace atdn {

namespace {
__device__ __forceinline__ float mish_grad(float x) {
  if (x > 20.0f) return 1.0f;
  const float sp = log1pf(expf(x));
  const float t = tanhf(sp);
  return t + x * (1.0f - t * t) * sigmoidf_(x);
}
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
constexpr int kRedThreads = 256;
constexpr long kPixPerBlock = 16384;  // pixels one reduction block walks over
}  // namespace

// ------------------------------------------------------------------------------------------------ weight packing
__global__ void pack_row_kernel(const float* __restrict__ w, int N, int Cin, int Cpix, int KH, int KW, int transposed,
                                int rows, int ldr, float* __restrict__ dst) {
  const long total = (long)rows * KH * ldr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / ((long)KH * ldr));
    const int rem = (int)(i - (long)row * KH * ldr);
    const int ky = rem / ldr, q = rem - ky * ldr;
    const int kx = q / Cpix, ch = q - kx * Cpix;
    float v = 0.f;
    if (kx < KW) {
      if (!transposed) {
        if (ch < Cin) v = w[(((long)row * Cin + ch) * KH + ky) * KW + kx];
      } else {
        if (ch < N) v = w[(((long)ch * Cin + row) * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)];
      }
    }
    dst[i] = v;
  }
}
void launch_pack_row(const float* w, int N, int Cin, int Cpix, int KH, int KW, bool transposed, float* dst, hipStream_t st) {
  const int rows = transposed ? Cin : N;
  const int ldr = round_up(KW * Cpix, 32);
  const long total = (long)rows * KH * ldr;
  hipLaunchKernelGGL(pack_row_kernel, dim3((unsigned)std::min<long>(cdivl(total, 256), 1024)), dim3(256), 0, st, w, N, Cin,
                     Cpix, KH, KW, transposed ? 1 : 0, rows, ldr, dst);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ BatchNorm (train)
int bn_partial_blocks(long P) { return (int)cdivl(P, kPixPerBlock); }

// v1, v2 per element via F; block (blockIdx.x, group blockIdx.y) walks kPixPerBlock pixels, thread = one float4
template <class F>
__global__ __launch_bounds__(kRedThreads) void reduce2_kernel(long P, int nblk, float* __restrict__ part, F f) {
  const int g = blockIdx.y, blk = blockIdx.x;
  const int quad = threadIdx.x & 3, lane_pix = threadIdx.x >> 2;
  const long p0 = (long)blk * kPixPerBlock, p1 = min(p0 + kPixPerBlock, P);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (long p = p0 + lane_pix; p < p1; p += kRedThreads / 4) f((long)g * P + p, g, quad, s1, s2);
  __shared__ float red[2][kRedThreads][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 32) {  // 2 quantities x 16 channels
    const int which = threadIdx.x >> 4, ch = threadIdx.x & 15;
    float acc = 0.f;
    for (int t = (ch >> 2); t < kRedThreads; t += 4) acc += red[which][t][ch & 3];
    part[(((long)g * nblk + blk) * 2 + which) * 16 + ch] = acc;
  }
}

struct StatsFwd {
  const float4* z; int mish;
  __device__ __forceinline__ void operator()(long pix, int, int quad, float* s1, float* s2) const {
    const float4 v = z[pix * 4 + quad];
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float x = mish ? mishf_(a[e]) : a[e]; s1[e] += x; s2[e] += x * x; }
  }
};
void launch_bn_stats(const float* z, int G, long P, bool mish, float* part, hipStream_t st) {
  const int nblk = bn_partial_blocks(P);
  hipLaunchKernelGGL((reduce2_kernel<StatsFwd>), dim3(nblk, G), dim3(kRedThreads), 0, st, P, nblk, part,
                     StatsFwd{reinterpret_cast<const float4*>(z), mish ? 1 : 0});
  ATDN_HIP(hipGetLastError());
}

__global__ void bn_finalize_kernel(const float* __restrict__ part, int G, int nblk, double invP, double unbias,
                                   float* __restrict__ rm, float* __restrict__ rv, float* __restrict__ mean,
                                   float* __restrict__ rstd) {
  const int ch = threadIdx.x;
  if (ch >= 16) return;
  float m_run = rm[ch], v_run = rv[ch];
  for (int g = 0; g < G; ++g) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
      s1 += (double)part[(((long)g * nblk + b) * 2 + 0) * 16 + ch];
      s2 += (double)part[(((long)g * nblk + b) * 2 + 1) * 16 + ch];
    }
    const double mu = s1 * invP;
    double var = s2 * invP - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[g * 16 + ch] = (float)mu;
    rstd[g * 16 + ch] = (float)(1.0 / sqrt(var + 1e-5));
    m_run = 0.9f * m_run + 0.1f * (float)mu;                 // torch: running = (1-momentum)*running + momentum*batch
    v_run = 0.9f * v_run + 0.1f * (float)(var * unbias);
  }
  rm[ch] = m_run;
  rv[ch] = v_run;
}
void launch_bn_finalize(const float* part, int G, long P, float* running_mean, float* running_var, float* mean, float* rstd,
                        hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, st, part, G, bn_partial_blocks(P), 1.0 / (double)P,
                     P > 1 ? (double)P / (double)(P - 1) : 1.0, running_mean, running_var, mean, rstd);
  ATDN_HIP(hipGetLastError());
}

__global__ void bn_apply_kernel(const float4* __restrict__ z, long P, int mish, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float4* __restrict__ add, float4* __restrict__ y,
                                long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int quad = (int)(i & 3);
  const int g = (int)((i >> 2) / P);
  const float4 v = z[i];
  float a[4] = {v.x, v.y, v.z, v.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = quad * 4 + e;
    const float x = mish ? mishf_(a[e]) : a[e];
    o[e] = (x - mean[g * 16 + ch]) * rstd[g * 16 + ch] * gamma[ch] + beta[ch];
  }
  if (add) { const float4 r = add[i]; o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
  y[i] = make_float4(o[0], o[1], o[2], o[3]);
}
void launch_bn_apply(const float* z, int G, long P, bool mish, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const float* add, float* y, hipStream_t st) {
  const long total4 = (long)G * P * 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)cdivl(total4, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(z), P, mish ? 1 : 0, mean, rstd, gamma, beta,
                     reinterpret_cast<const float4*>(add), reinterpret_cast<float4*>(y), total4);
  ATDN_HIP(hipGetLastError());
}

struct StatsBwd {
  const float4* dy; const float4* z; int mish; const float* mean; const float* rstd;
  __device__ __forceinline__ void operator()(long pix, int g, int quad, float* s1, float* s2) const {
    const float4 d = dy[pix * 4 + quad], v = z[pix * 4 + quad];
    const float dd[4] = {d.x, d.y, d.z, d.w}, a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = quad * 4 + e;
      const float x = mish ? mishf_(a[e]) : a[e];
      const float xh = (x - mean[g * 16 + ch]) * rstd[g * 16 + ch];
      s1[e] += dd[e];
      s2[e] += dd[e] * xh;
    }
  }
};
void launch_bn_bwd_stats(const float* dy, const float* z, int G, long P, bool mish, const float* mean, const float* rstd,
                         float* part, hipStream_t st) {
  const int nblk = bn_partial_blocks(P);
  hipLaunchKernelGGL((reduce2_kernel<StatsBwd>), dim3(nblk, G), dim3(kRedThreads), 0, st, P, nblk, part,
                     StatsBwd{reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(z), mish ? 1 : 0, mean, rstd});
  ATDN_HIP(hipGetLastError());
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int G, int nblk, float* __restrict__ sums,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int ch = threadIdx.x;
  if (ch >= 16) return;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < G; ++g) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
      s1 += (double)part[(((long)g * nblk + b) * 2 + 0) * 16 + ch];
      s2 += (double)part[(((long)g * nblk + b) * 2 + 1) * 16 + ch];
    }
    sums[(g * 2 + 0) * 16 + ch] = (float)s1;
    sums[(g * 2 + 1) * 16 + ch] = (float)s2;
    tb += s1;
    tg += s2;
  }
  dgamma[ch] += (float)tg;
  dbeta[ch] += (float)tb;
}
void launch_bn_bwd_finalize(const float* part, int G, long P, float* sums, float* dgamma, float* dbeta, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(64), 0, st, part, G, bn_partial_blocks(P), sums, dgamma, dbeta);
  ATDN_HIP(hipGetLastError());
}

// dz and the per-block sums of dz (bias gradient of the convolution that produced z)
__global__ __launch_bounds__(kRedThreads) void bn_bwd_apply_kernel(const float4* __restrict__ dy, const float4* __restrict__ z,
                                                                   long P, int nblk, int mish, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ sums, float invP,
                                                                   float4* __restrict__ dz, float* __restrict__ part_db) {
  const int g = blockIdx.y, blk = blockIdx.x;
  const int quad = threadIdx.x & 3, lane_pix = threadIdx.x >> 2;
  const long p0 = (long)blk * kPixPerBlock, p1 = min(p0 + kPixPerBlock, P);
  float mu[4], rs[4], ga[4], m1[4], m2[4], sdb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = quad * 4 + e;
    mu[e] = mean[g * 16 + ch]; rs[e] = rstd[g * 16 + ch]; ga[e] = gamma[ch];
    m1[e] = sums[(g * 2 + 0) * 16 + ch] * invP;
    m2[e] = sums[(g * 2 + 1) * 16 + ch] * invP;
  }
  for (long p = p0 + lane_pix; p < p1; p += kRedThreads / 4) {
    const long i = ((long)g * P + p) * 4 + quad;
    const float4 d = dy[i], v = z[i];
    const float dd[4] = {d.x, d.y, d.z, d.w}, a[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = mish ? mishf_(a[e]) : a[e];
      const float xh = (x - mu[e]) * rs[e];
      const float da = ga[e] * rs[e] * (dd[e] - m1[e] - xh * m2[e]);
      o[e] = mish ? da * mish_grad(a[e]) : da;
      sdb[e] += o[e];
    }
    dz[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
  __shared__ float red[kRedThreads][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = sdb[e];
  __syncthreads();
  if (threadIdx.x < 16) {
    const int ch = threadIdx.x;
    float acc = 0.f;
    for (int t = (ch >> 2); t < kRedThreads; t += 4) acc += red[t][ch & 3];
    part_db[((long)g * nblk + blk) * 16 + ch] = acc;
  }
}
void launch_bn_bwd_apply(const float* dy, const float* z, int G, long P, bool mish, const float* mean, const float* rstd,
                         const float* gamma, const float* sums, float* dz, float* part_db, hipStream_t st) {
  const int nblk = bn_partial_blocks(P);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk, G), dim3(kRedThreads), 0, st, reinterpret_cast<const float4*>(dy),
                     reinterpret_cast<const float4*>(z), P, nblk, mish ? 1 : 0, mean, rstd, gamma, sums, 1.0f / (float)P,
                     reinterpret_cast<float4*>(dz), part_db);
  ATDN_HIP(hipGetLastError());
}

__global__ void sum_partials16_kernel(const float* __restrict__ part, long rows, float* __restrict__ out) {
  const int ch = threadIdx.x;
  if (ch >= 16) return;
  double s = 0.0;
  for (long r = 0; r < rows; ++r) s += (double)part[r * 16 + ch];
  out[ch] += (float)s;
}
void launch_sum_partials16(const float* part, long rows, float* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_partials16_kernel, dim3(1), dim3(64), 0, st, part, rows, out);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ transposed conv
__global__ void zero_stuff_kernel(const float4* __restrict__ dz, int Ho, int Wo, int stride, int Hs, int Ws,
                                  float4* __restrict__ D, long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int quad = (int)(i & 3);
  long pix = i >> 2;
  const int x = (int)(pix % Ws); pix /= Ws;
  const int y = (int)(pix % Hs);
  const long img = pix / Hs;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y % stride == 0 && x % stride == 0) {
    const int oy = y / stride, ox = x / stride;
    if (oy < Ho && ox < Wo) v = dz[((img * Ho + oy) * Wo + ox) * 4 + quad];
  }
  D[i] = v;
}
void launch_zero_stuff(const float* dz, int nimg, int Ho, int Wo, int stride, int Hs, int Ws, float* D, hipStream_t st) {
  const long total4 = (long)nimg * Hs * Ws * 4;
  hipLaunchKernelGGL(zero_stuff_kernel, dim3((unsigned)cdivl(total4, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(dz), Ho, Wo, stride, Hs, Ws, reinterpret_cast<float4*>(D), total4);
  ATDN_HIP(hipGetLastError());
}

__global__ void add_inplace_kernel(float4* __restrict__ a, const float4* __restrict__ b, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 x = a[i];
  const float4 y = b[i];
  x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
  a[i] = x;
}
void launch_add_inplace(float* a, const float* b, long n, hipStream_t st) {
  ATDN_CHECK(n % 4 == 0, "add_inplace: length must be a multiple of 4");
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)cdivl(n / 4, 256)), dim3(256), 0, st, reinterpret_cast<float4*>(a),
                     reinterpret_cast<const float4*>(b), n / 4);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ weight gradients
namespace {
constexpr int WG_R = 4;                // output rows staged per step; the column count is chosen to fit LDS
constexpr int WG_BLOCKS = 1024;       // persistent blocks; partials are [output][block]
constexpr int WG_MAXO = 9;            // outputs per thread (3x3x16x16 = 2304 = 9 x 256)
}
long wgrad_scratch_floats(int, int, int Cin, int KH, int KW) { return (long)WG_BLOCKS * KH * KW * 16 * Cin; }

// Thread t owns outputs o = t + 256*k (k < WG_MAXO), o = (tap*16 + n)*Cin + c: for Cin = 16 a thread keeps one (n, c)
// pair for all taps, a wave reads 16 consecutive channels of one patch pixel (conflict-free, broadcast over n).
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, int Cpix, int Cin, int nimg, int H, int W,
                                                         const float* __restrict__ dz, int Ho, int Wo, int KH, int KW,
                                                         int stride, int pad, int WG_TW, float* __restrict__ scratch) {
  extern __shared__ float lds[];
  const int total = KH * KW * 16 * Cin;
  const int PH = (WG_R - 1) * stride + KH, PWp = (WG_TW - 1) * stride + KW;
  float* xs = lds;                          // [PH][PWp][Cpix]
  float* ds = lds + PH * PWp * Cpix;        // [WG_R][WG_TW][16]
  int off_x[WG_MAXO], n_o[WG_MAXO];
  float acc[WG_MAXO];
#pragma unroll
  for (int k = 0; k < WG_MAXO; ++k) {
    const int o = threadIdx.x + 256 * k;
    acc[k] = 0.f;
    if (o < total) {
      const int c = o % Cin, n = (o / Cin) & 15, tap = o / (Cin * 16);
      const int ky = tap / KW, kx = tap - ky * KW;
      off_x[k] = (ky * PWp + kx) * Cpix + c;
      n_o[k] = n;
    } else {
      off_x[k] = -1;
      n_o[k] = 0;
    }
  }
  const int rblocks = cdiv_dev(Ho, WG_R), cblocks = cdiv_dev(Wo, WG_TW);
  const long items = (long)nimg * rblocks * cblocks;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int cb = (int)(it % cblocks);
    const int rb = (int)((it / cblocks) % rblocks);
    const int img = (int)(it / ((long)cblocks * rblocks));
    const int oy0 = rb * WG_R, ox0 = cb * WG_TW;
    const int iy0 = oy0 * stride - pad, ix0 = ox0 * stride - pad;
    __syncthreads();
    for (int i = threadIdx.x; i < PH * PWp * Cpix; i += 256) {
      const int ch = i % Cpix, px = (i / Cpix) % PWp, py = i / (Cpix * PWp);
      const int iy = iy0 + py, ix = ix0 + px;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((long)img * H + iy) * W + ix) * Cpix + ch];
      xs[i] = v;
    }
    for (int i = threadIdx.x; i < WG_R * WG_TW * 16; i += 256) {
      const int ch = i & 15, col = (i >> 4) % WG_TW, r = i / (16 * WG_TW);
      const int oy = oy0 + r, ox = ox0 + col;
      float v = 0.f;
      if (oy < Ho && ox < Wo) v = dz[(((long)img * Ho + oy) * Wo + ox) * 16 + ch];
      ds[i] = v;
    }
    __syncthreads();
    for (int r = 0; r < WG_R; ++r)
      for (int col = 0; col < WG_TW; ++col) {
        const float* dp = ds + (r * WG_TW + col) * 16;
        const float* xp = xs + (r * stride * PWp + col * stride) * Cpix;
#pragma unroll
        for (int k = 0; k < WG_MAXO; ++k)
          if (off_x[k] >= 0) acc[k] += dp[n_o[k]] * xp[off_x[k]];
      }
  }
#pragma unroll
  for (int k = 0; k < WG_MAXO; ++k) {
    const int o = threadIdx.x + 256 * k;
    if (o < total) scratch[(long)o * gridDim.x + blockIdx.x] = acc[k];
  }
}

// one wave per output: sum its WG_BLOCKS partials, add into dW (OIHW)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ scratch, int nblk, int Cin, int KH,
                                                                int KW, float* __restrict__ dW) {
  const int total = KH * KW * 16 * Cin;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= total) return;
  float s = 0.f;
  for (int b = lane; b < nblk; b += 64) s += scratch[(long)o * nblk + b];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) {
    const int c = o % Cin, n = (o / Cin) & 15, tap = o / (Cin * 16);
    dW[((long)n * Cin + c) * KH * KW + tap] += s;
  }
}

void launch_conv_wgrad(const float* x, int Cpix, int Cin, int nimg, int H, int W, const float* dz, int Ho, int Wo, int KH,
                       int KW, int stride, int pad, float* scratch, float* dW, hipStream_t st) {
  const int total = KH * KW * 16 * Cin;
  ATDN_CHECK(total <= 256 * WG_MAXO, "conv_wgrad: too many weights per output channel block");
  const int PH = (WG_R - 1) * stride + KH;
  int WG_TW = 32;
  size_t lds = 0;
  for (;; WG_TW /= 2) {
    const int PWp = (WG_TW - 1) * stride + KW;
    lds = (size_t)(PH * PWp * Cpix + WG_R * WG_TW * 16) * sizeof(float);
    if (lds <= 48 * 1024 || WG_TW == 4) break;
  }
  ATDN_CHECK(lds <= 64 * 1024, "conv_wgrad: staging tile exceeds the LDS budget");
  const long items = (long)nimg * cdiv(Ho, WG_R) * cdiv(Wo, WG_TW);
  const int nblk = (int)std::min<long>(items, WG_BLOCKS);
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(nblk), dim3(256), lds, st, x, Cpix, Cin, nimg, H, W, dz, Ho, Wo, KH, KW, stride,
                     pad, WG_TW, scratch);
  ATDN_HIP(hipGetLastError());
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(total, 4)), dim3(256), 0, st, scratch, nblk, Cin, KH, KW, dW);
  ATDN_HIP(hipGetLastError());
}

// depthwise 1x1 gradients: 4 sums (dw0, dw1, db0, db1)
__global__ __launch_bounds__(256) void dw_grad_kernel(const float* __restrict__ flow, const float4* __restrict__ dx0, int nimg,
                                                      long HW, float sx, float sy, float* __restrict__ scratch) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const long total = (long)nimg * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long img = i / HW, p = i - img * HW;
    const float4 d = dx0[i];
    const float* f = flow + img * 2 * HW + p;
    s[0] += d.x * (f[0] / sx);
    s[1] += d.y * (f[HW] / sy);
    s[2] += d.x;
    s[3] += d.y;
  }
  __shared__ float red[256][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (threadIdx.x < 4) {
    float a = 0.f;
    for (int t = 0; t < 256; ++t) a += red[t][threadIdx.x];
    scratch[blockIdx.x * 4 + threadIdx.x] = a;
  }
}
__global__ void dw_grad_reduce_kernel(const float* __restrict__ scratch, int nblk, float* __restrict__ dw, float* __restrict__ db) {
  if (threadIdx.x >= 4) return;
  double a = 0.0;
  for (int b = 0; b < nblk; ++b) a += (double)scratch[b * 4 + threadIdx.x];
  if (threadIdx.x < 2) dw[threadIdx.x] += (float)a; else db[threadIdx.x - 2] += (float)a;
}
void launch_dw_grad(const float* flow, const float* dx0, int nimg, long HW, float* scratch, float* dw, float* db, hipStream_t st) {
  const int nblk = 512;
  hipLaunchKernelGGL(dw_grad_kernel, dim3(nblk), dim3(256), 0, st, flow, reinterpret_cast<const float4*>(dx0), nimg, HW,
                     58.1837f, 17.7647f, scratch);
  ATDN_HIP(hipGetLastError());
  hipLaunchKernelGGL(dw_grad_reduce_kernel, dim3(1), dim3(64), 0, st, scratch, nblk, dw, db);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ small dense algebra
// 64x64 output tile, 16x16 threads x 4x4 values, K step 16; op(A) is M x K, op(B) is K x N
__global__ __launch_bounds__(256) void gemm_kernel(int transA, int transB, int M, int N, int K, const float* __restrict__ A,
                                                   int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                   int ldc, float beta, const float* __restrict__ bias) {
  __shared__ float As[16][64 + 1], Bs[16][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int kk = i & 15, mm = i >> 4;   // consecutive threads walk K for the non-transposed A (rows contiguous)
      const int m = m0 + mm, k = k0 + kk;
      float v = 0.f;
      if (m < M && k < K) v = transA ? A[(long)k * lda + m] : A[(long)m * lda + k];
      As[kk][mm] = v;
    }
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int nn = i & 63, kk = i >> 6;
      const int n = n0 + nn, k = k0 + kk;
      float v = 0.f;
      if (n < N && k < K) v = transB ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) {
        float v = acc[i][j] + (bias ? bias[n] : 0.f);
        if (beta != 0.f) v += beta * C[(long)m * ldc + n];
        C[(long)m * ldc + n] = v;
      }
    }
}
void launch_gemm(bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                 int ldc, float beta, const float* bias, hipStream_t st) {
  hipLaunchKernelGGL(gemm_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, st, transA ? 1 : 0, transB ? 1 : 0, M, N, K, A,
                     lda, B, ldb, C, ldc, beta, bias);
  ATDN_HIP(hipGetLastError());
}

__global__ void colsum_kernel(const float* __restrict__ X, int rows, int cols, int ld, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += X[(long)r * ld + c];
  out[c] += s;
}
void launch_colsum(const float* X, int rows, int cols, int ld, float* out, hipStream_t st) {
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, X, rows, cols, ld, out);
  ATDN_HIP(hipGetLastError());
}

__global__ void mish_fwd_kernel(const float* __restrict__ z, float* __restrict__ a, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = mishf_(z[i]);
}
void launch_mish_fwd(const float* z, float* a, long n, hipStream_t st) {
  hipLaunchKernelGGL(mish_fwd_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, z, a, n);
  ATDN_HIP(hipGetLastError());
}
__global__ void mish_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = dy[i] * mish_grad(z[i]);
}
void launch_mish_bwd(const float* dy, const float* z, float* dz, long n, hipStream_t st) {
  hipLaunchKernelGGL(mish_bwd_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, dy, z, dz, n);
  ATDN_HIP(hipGetLastError());
}

__global__ void permute_p16_kernel(const float* __restrict__ src, int P, int to_chw, float* __restrict__ dst, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long img = i / (16L * P);
  const int r = (int)(i - img * 16L * P);
  if (to_chw) {  // dst index r = c*P + p
    const int c = r / P, p = r - c * P;
    dst[i] = src[img * 16L * P + (long)p * 16 + c];
  } else {       // dst index r = p*16 + c
    const int p = r >> 4, c = r & 15;
    dst[i] = src[img * 16L * P + (long)c * P + p];
  }
}
void launch_nhwc_to_chw(const float* src, int nimg, int P, float* dst, hipStream_t st) {
  const long total = (long)nimg * 16 * P;
  hipLaunchKernelGGL(permute_p16_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, src, P, 1, dst, total);
  ATDN_HIP(hipGetLastError());
}
void launch_chw_to_nhwc(const float* src, int nimg, int P, float* dst, hipStream_t st) {
  const long total = (long)nimg * 16 * P;
  hipLaunchKernelGGL(permute_p16_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, src, P, 0, dst, total);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ LSTM cell
__global__ void lstm_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ c_in, int B, float* __restrict__ act,
                                float* __restrict__ c_out, float* __restrict__ tanhc, float* __restrict__ h_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 512) return;
  const int b = i >> 9, j = i & 511;
  const float* p = pre + (long)b * 2048;
  const float ig = sigmoidf_(p[j]), fg = sigmoidf_(p[512 + j]), gg = tanhf(p[1024 + j]), og = sigmoidf_(p[1536 + j]);
  float* a = act + (long)b * 2048;
  a[j] = ig; a[512 + j] = fg; a[1024 + j] = gg; a[1536 + j] = og;
  const float c = fg * c_in[i] + ig * gg;
  const float tc = tanhf(c);
  c_out[i] = c;
  tanhc[i] = tc;
  h_out[i] = og * tc;
}
void launch_lstm_fwd(const float* pre, const float* c_in, int B, float* act, float* c_out, float* tanhc, float* h_out,
                     hipStream_t st) {
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(cdiv(B * 512, 256)), dim3(256), 0, st, pre, c_in, B, act, c_out, tanhc, h_out);
  ATDN_HIP(hipGetLastError());
}
__global__ void lstm_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc_out, const float* __restrict__ act,
                                const float* __restrict__ c_in, const float* __restrict__ tanhc, int B,
                                float* __restrict__ dpre, float* __restrict__ dc_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 512) return;
  const int b = i >> 9, j = i & 511;
  const float* a = act + (long)b * 2048;
  const float ig = a[j], fg = a[512 + j], gg = a[1024 + j], og = a[1536 + j];
  const float tc = tanhc[i];
  const float dhv = dh[i];
  const float dc = (dc_out ? dc_out[i] : 0.f) + dhv * og * (1.f - tc * tc);
  float* d = dpre + (long)b * 2048;
  d[j] = dc * gg * ig * (1.f - ig);
  d[512 + j] = dc * c_in[i] * fg * (1.f - fg);
  d[1024 + j] = dc * ig * (1.f - gg * gg);
  d[1536 + j] = dhv * tc * og * (1.f - og);
  dc_in[i] = dc * fg;
}
void launch_lstm_bwd(const float* dh, const float* dc_out, const float* act, const float* c_in, const float* tanhc, int B,
                     float* dpre, float* dc_in, hipStream_t st) {
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(cdiv(B * 512, 256)), dim3(256), 0, st, dh, dc_out, act, c_in, tanhc, B, dpre, dc_in);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ loss, optimiser
// rows r = t*B + b (step-major, as the trainer stores its activations); targets are [B][T][3] like the reference's
__global__ void clvo_loss_kernel(const float* __restrict__ pr, const float* __restrict__ pt, const float* __restrict__ tr_,
                                 const float* __restrict__ tt, int B, int T, float* __restrict__ loss, float* __restrict__ d_rot,
                                 float* __restrict__ d_tr) {
  __shared__ double red[256];
  double s = 0.0;
  const float invB = 1.0f / (float)B;
  for (int i = threadIdx.x; i < B * T * 3; i += blockDim.x) {
    const int e = i % 3, r = i / 3;
    const int t = r / B, b = r - t * B;
    const long ti = ((long)b * T + t) * 3 + e;
    const float dr = pr[i] - tr_[ti], dt = pt[i] - tt[ti];
    s += (double)(1.0f * dt * dt) + (double)(100.0f * dr * dr);
    d_rot[i] = 2.0f * 100.0f * dr * invB;
    d_tr[i] = 2.0f * 1.0f * dt * invB;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int t = 0; t < blockDim.x; ++t) a += red[t];
    loss[0] = (float)(a / (double)B);
  }
}
void launch_clvo_loss(const float* pred_rot, const float* pred_tr, const float* true_rot, const float* true_tr, int B, int T,
                      float* loss, float* d_rot, float* d_tr, hipStream_t st) {
  hipLaunchKernelGGL(clvo_loss_kernel, dim3(1), dim3(256), 0, st, pred_rot, pred_tr, true_rot, true_tr, B, T, loss, d_rot, d_tr);
  ATDN_HIP(hipGetLastError());
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float wd, float eps, float b1, float b2, float bc1, float sqrt_bc2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gr = g[i];
  float pv = p[i] * (1.f - lr * wd);
  const float mv = b1 * m[i] + (1.f - b1) * gr;
  const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
  m[i] = mv;
  v[i] = vv;
  const float denom = sqrtf(vv) / sqrt_bc2 + eps;
  pv -= (lr / bc1) * (mv / denom);
  p[i] = pv;
}
void launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float wd, float eps, float beta1, float beta2,
                  int t, hipStream_t st) {
  const double bc1 = 1.0 - std::pow((double)beta1, (double)t), bc2 = 1.0 - std::pow((double)beta2, (double)t);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, p, g, m, v, n, lr, wd, eps, beta1, beta2,
                     (float)bc1, (float)std::sqrt(bc2));
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
