// See train_kernels.h. Everything here is HBM- or latency-bound bookkeeping around the convolutions; kernels are
// written for coalesced 16-byte accesses (NHWC16 maps: one float4 per thread) and deterministic reductions
// (per-block partials + a second pass, no floating-point atomics).
#include "train_kernels.h"

#include <cstdint>
#include <cstdlib>
#include "common.h"
#include <algorithm>
#include <cmath>

namespace atdn {

namespace {
// Mish and its derivative from ONE exponential: with n = e^x, tanh(softplus(x)) = t/(t+2), t = n(n+2).
// (The library formula x*tanh(log1p(exp(x))) costs ~40 instructions per element and made the BatchNorm passes
// ALU-bound at 0.9 TB/s; the two forms agree to a few ulp.)
struct MishVal { float y, dy; };
__device__ __forceinline__ MishVal mish_both(float x) {
  if (x > 20.0f) return {x, 1.0f};
  // (round 5: v_rcp_f32 — 1 ulp — instead of the two IEEE divisions, ~10 instructions each: the statistics taken inside the
  // convolution kernels pay for every vector instruction of their epilogue)
  const float n = __expf(x);
  const float t = n * (n + 2.0f);
  const float th = t * __builtin_amdgcn_rcpf(t + 2.0f);
  const float sg = n * __builtin_amdgcn_rcpf(1.0f + n);
  return {x * th, th + x * (1.0f - th * th) * sg};
}
__device__ __forceinline__ float mish_fast(float x) { return mish_both(x).y; }
__device__ __forceinline__ float mish_grad(float x) { return mish_both(x).dy; }
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
  constexpr int kStep = kRedThreads / 4;
  long p = p0 + lane_pix;
  for (; p + 3 * kStep < p1; p += 4 * kStep) {   // four independent loads in flight per thread
    typename F::Elem e0 = f.load((long)g * P + p, quad), e1 = f.load((long)g * P + p + kStep, quad),
                     e2 = f.load((long)g * P + p + 2 * kStep, quad), e3 = f.load((long)g * P + p + 3 * kStep, quad);
    f.acc(e0, g, quad, s1, s2); f.acc(e1, g, quad, s1, s2); f.acc(e2, g, quad, s1, s2); f.acc(e3, g, quad, s1, s2);
  }
  for (; p < p1; p += kStep) f.acc(f.load((long)g * P + p, quad), g, quad, s1, s2);
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
  typedef float4 Elem;
  __device__ __forceinline__ Elem load(long pix, int quad) const { return z[pix * 4 + quad]; }
  __device__ __forceinline__ void acc(const Elem& v, int, int quad, float* s1, float* s2) const {
    const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float x = mish ? mish_fast(a[e]) : a[e]; s1[e] += x; s2[e] += x * x; }
  }
};
void launch_bn_stats(const float* z, int G, long P, bool mish, float* part, hipStream_t st) {
  const int nblk = bn_partial_blocks(P);
  hipLaunchKernelGGL((reduce2_kernel<StatsFwd>), dim3(nblk, G), dim3(kRedThreads), 0, st, P, nblk, part,
                     StatsFwd{reinterpret_cast<const float4*>(z), mish ? 1 : 0});
  ATDN_HIP(hipGetLastError());
}

// 256 threads = 16 slices x 16 channels: a slice sums every 16th partial in double, thread ch < 16 adds the slices in order
__device__ __forceinline__ void sum_partials2(const float* __restrict__ part_g, int nblk, double (*r)[16][16], double& s1,
                                              double& s2) {
  const int ch = threadIdx.x & 15, sl = threadIdx.x >> 4;
  double a = 0.0, b = 0.0;
  for (int k = sl; k < nblk; k += 16) {
    a += (double)part_g[((long)k * 2 + 0) * 16 + ch];
    b += (double)part_g[((long)k * 2 + 1) * 16 + ch];
  }
  __syncthreads();   // r is reused between groups
  r[0][sl][ch] = a; r[1][sl][ch] = b;
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
  if (threadIdx.x < 16)
    for (int k = 0; k < 16; ++k) { s1 += r[0][k][ch]; s2 += r[1][k][ch]; }
}

// One block per group sums its partial rows (round 5: up to 1,024 of them when a convolution kernel took the statistics — the
// groups one after the other in ONE block cost 70 us per layer); a second, 16-thread launch walks the running statistics through
// the groups IN ORDER, as G successive forward() calls would. (Not a last-block-finishes counter: a device-scope release on this
// chip writes back an XCD's whole L2, kernels.hip.)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, double invP, double unbias,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ var_g) {
  __shared__ double r[2][16][16];
  const int ch = threadIdx.x & 15, g = blockIdx.x;
  double s1, s2;
  sum_partials2(part + (long)g * nblk * 32, nblk, r, s1, s2);
  if (threadIdx.x < 16) {
    const double mu = s1 * invP;
    double var = s2 * invP - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[g * 16 + ch] = (float)mu;
    rstd[g * 16 + ch] = (float)(1.0 / sqrt(var + 1e-5));
    var_g[g * 16 + ch] = (float)(var * unbias);
  }
}
__global__ void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var_g, int G, float* __restrict__ rm,
                                  float* __restrict__ rv) {
  const int ch = threadIdx.x;
  float m_run = rm[ch], v_run = rv[ch];
  for (int g = 0; g < G; ++g) {
    m_run = 0.9f * m_run + 0.1f * mean[g * 16 + ch];                 // torch: running = (1-momentum)*running + momentum*batch
    v_run = 0.9f * v_run + 0.1f * var_g[g * 16 + ch];
  }
  rm[ch] = m_run; rv[ch] = v_run;
}
void launch_bn_finalize_rows(const float* part, int G, int rows, long P, float* running_mean, float* running_var, float* mean,
                             float* rstd, float* var_scratch, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(G), dim3(256), 0, st, part, rows, 1.0 / (double)P,
                     P > 1 ? (double)P / (double)(P - 1) : 1.0, mean, rstd, var_scratch);
  ATDN_HIP(hipGetLastError());
  hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(16), 0, st, mean, var_scratch, G, running_mean, running_var);
  ATDN_HIP(hipGetLastError());
}
void launch_bn_finalize(const float* part, int G, long P, float* running_mean, float* running_var, float* mean, float* rstd,
                        float* var_scratch, hipStream_t st) {
  launch_bn_finalize_rows(part, G, bn_partial_blocks(P), P, running_mean, running_var, mean, rstd, var_scratch, st);
}

// (round 5: one group per blockIdx.y and a strip of pixels per block, four float4 — eight with the residual — in flight per thread and
// the channel constants in registers; the one-float4-per-thread form with its 64-bit division per thread ran at 2.6 TB/s on the
// largest map)
constexpr long kApplyPix = 4096;   // pixels per block
__global__ __launch_bounds__(256) void bn_apply_kernel(const float4* __restrict__ z, long P, int mish, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float4* __restrict__ add,
                                                       float4* __restrict__ y, float* __restrict__ next_part) {
  const int g = blockIdx.y;
  const int quad = threadIdx.x & 3, lane_pix = threadIdx.x >> 2;
  const long p0 = (long)blockIdx.x * kApplyPix, p1 = min(p0 + kApplyPix, P);
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = quad * 4 + e;
    mu[e] = mean[g * 16 + ch]; rs[e] = rstd[g * 16 + ch]; ga[e] = gamma[ch]; be[e] = beta[ch];
  }
  float n1[4] = {0.f, 0.f, 0.f, 0.f}, n2[4] = {0.f, 0.f, 0.f, 0.f};   // next_part: sums of Mish(y), Mish(y)^2
  auto one = [&](const float4 v, const float4 r) {
    const float a[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = mish ? mish_fast(a[e]) : a[e];
      o[e] = (x - mu[e]) * rs[e] * ga[e] + be[e];
      o[e] += rr[e];   // (zeros without a residual: x + 0 = x)
      if (next_part) { const float m = mish_fast(o[e]); n1[e] += m; n2[e] += m * m; }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
  };
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int kStep = 256 / 4;
  long p = p0 + lane_pix;
  for (; p + 3 * kStep < p1; p += 4 * kStep) {
    const long i0 = ((long)g * P + p) * 4 + quad, i1 = i0 + 4L * kStep, i2 = i0 + 8L * kStep, i3 = i0 + 12L * kStep;
    const float4 v0 = z[i0], v1 = z[i1], v2 = z[i2], v3 = z[i3];
    float4 r0 = zero, r1 = zero, r2 = zero, r3 = zero;
    if (add) { r0 = add[i0]; r1 = add[i1]; r2 = add[i2]; r3 = add[i3]; }
    y[i0] = one(v0, r0); y[i1] = one(v1, r1); y[i2] = one(v2, r2); y[i3] = one(v3, r3);
  }
  for (; p < p1; p += kStep) {
    const long i = ((long)g * P + p) * 4 + quad;
    y[i] = one(z[i], add ? add[i] : zero);
  }
  if (next_part) {   // the block's partial row, reduced like reduce2_kernel's
    __shared__ float red[2][256][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = n1[e]; red[1][threadIdx.x][e] = n2[e]; }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int which = threadIdx.x >> 4, ch = threadIdx.x & 15;
      float acc = 0.f;
      for (int t = (ch >> 2); t < 256; t += 4) acc += red[which][t][ch & 3];
      next_part[(((long)g * gridDim.x + blockIdx.x) * 2 + which) * 16 + ch] = acc;
    }
  }
}
int bn_apply_partial_rows(long P) { return (int)cdivl(P, kApplyPix); }
void launch_bn_apply(const float* z, int G, long P, bool mish, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const float* add, float* y, hipStream_t st, float* next_part) {
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)cdivl(P, kApplyPix), G), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(z), P, mish ? 1 : 0, mean, rstd, gamma, beta,
                     reinterpret_cast<const float4*>(add), reinterpret_cast<float4*>(y), next_part);
  ATDN_HIP(hipGetLastError());
}

struct StatsBwd {
  const float4* dy; const float4* z; int mish; const float* mean; const float* rstd;
  struct Elem { float4 d, v; };
  __device__ __forceinline__ Elem load(long pix, int quad) const { return {dy[pix * 4 + quad], z[pix * 4 + quad]}; }
  __device__ __forceinline__ void acc(const Elem& el, int g, int quad, float* s1, float* s2) const {
    const float4 d = el.d, v = el.v;
    const float dd[4] = {d.x, d.y, d.z, d.w}, a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = quad * 4 + e;
      const float x = mish ? mish_fast(a[e]) : a[e];
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

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int G, int nblk,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  __shared__ double r[2][16][16];
  const int ch = threadIdx.x & 15;
  const bool lead = threadIdx.x < 16;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < G; ++g) {
    double s1, s2;
    sum_partials2(part + (long)g * nblk * 32, nblk, r, s1, s2);
    if (lead) {
      sums[(g * 2 + 0) * 16 + ch] = (float)s1;
      sums[(g * 2 + 1) * 16 + ch] = (float)s2;
      tb += s1;
      tg += s2;
    }
  }
  if (lead) { dgamma[ch] += (float)tg; dbeta[ch] += (float)tb; }
}
void launch_bn_bwd_finalize(const float* part, int G, long P, float* sums, float* dgamma, float* dbeta, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, part, G, bn_partial_blocks(P), sums, dgamma, dbeta);
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
  auto one = [&](long i, const float4 d, const float4 v) {
    const float dd[4] = {d.x, d.y, d.z, d.w}, a[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      MishVal mv = {a[e], 1.0f};
      if (mish) mv = mish_both(a[e]);
      const float xh = (mv.y - mu[e]) * rs[e];
      const float da = ga[e] * rs[e] * (dd[e] - m1[e] - xh * m2[e]);
      o[e] = da * mv.dy;
      sdb[e] += o[e];
    }
    dz[i] = make_float4(o[0], o[1], o[2], o[3]);
  };
  constexpr int kStep = kRedThreads / 4;
  long p = p0 + lane_pix;
  for (; p + 3 * kStep < p1; p += 4 * kStep) {   // eight independent loads in flight per thread
    const long i0 = ((long)g * P + p) * 4 + quad, i1 = i0 + 4L * kStep, i2 = i0 + 8L * kStep, i3 = i0 + 12L * kStep;
    const float4 d0 = dy[i0], v0 = z[i0], d1 = dy[i1], v1 = z[i1], d2 = dy[i2], v2 = z[i2], d3 = dy[i3], v3 = z[i3];
    one(i0, d0, v0); one(i1, d1, v1); one(i2, d2, v2); one(i3, d3, v3);
  }
  for (; p < p1; p += kStep) { const long i = ((long)g * P + p) * 4 + quad; one(i, dy[i], z[i]); }
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

__global__ __launch_bounds__(256) void sum_partials16_kernel(const float* __restrict__ part, long rows, float* __restrict__ out) {
  __shared__ double r[16][16];
  const int ch = threadIdx.x & 15, sl = threadIdx.x >> 4;
  double s = 0.0;
  for (long k = sl; k < rows; k += 16) s += (double)part[k * 16 + ch];
  r[sl][ch] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += r[k][ch];
    out[ch] += (float)t;
  }
}
void launch_sum_partials16(const float* part, long rows, float* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_partials16_kernel, dim3(1), dim3(256), 0, st, part, rows, out);
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
constexpr int WG_TW = 32;        // output columns staged per step
constexpr int WG_BLOCKS = 1024;  // persistent blocks; partials are [output][block]
constexpr int WG_MFMA_BLOCKS = 768, WG_SLOTS = WG_MFMA_BLOCKS * 4;   // MFMA variant: one partial per wave
constexpr int wg_nf4(int S, int KW) { return (3 * S + KW + 3) / 4; }   // float4 loads covering 4 output columns of one tap row
constexpr int wg_pwpad(int S, int KW) {
  int need = (WG_TW - 4) * S + wg_nf4(S, KW) * 4;
  int p = ((WG_TW - 1) * S + KW + 3) / 4 * 4;
  if (p < need) p = need;
  if ((p / 4) % 2 == 0) p += 4;   // odd multiple of 4 dwords: the 16 channel rows of a ds_read_b128 group hit 16 distinct bank quads
  return p;
}
}  // namespace
long wgrad_scratch_floats(int, int, int Cin, int KH, int KW) { return (long)WG_SLOTS * KH * KW * 16 * Cin; }

// Thread (n, c, g): output channel n, input channel c, and the kernel rows ky = g, g + GROUPS, ... with ALL kx of those
// rows (GROUPS = 256 / (16*CIN)). The staged input patch is channel-major in LDS, xs[py][c][px], so the 3*S + KW
// consecutive px a thread needs for 4 adjacent output columns are NF4 aligned ds_read_b128; the gradient tile is
// ds[r][n][col] (one b128 per 4 columns): 0.2 LDS loads per FMA instead of 1.1 with a pixel-major patch.
template <int S, int KH, int KW, int CIN, int R>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, int Cpix, int nimg, int H, int W,
                                                         const float* __restrict__ dz, int Ho, int Wo, int pad,
                                                         float* __restrict__ scratch) {
  constexpr int PAIRS = 16 * CIN, GROUPS = 256 / PAIRS, NKY = (KH + GROUPS - 1) / GROUPS;
  constexpr int PH = (R - 1) * S + KH, PWP = wg_pwpad(S, KW), NF4 = wg_nf4(S, KW);
  __shared__ __attribute__((aligned(16))) float xs[PH * CIN * PWP];
  __shared__ __attribute__((aligned(16))) float ds[R * 16 * WG_TW];
  const int pair = threadIdx.x % PAIRS, g = threadIdx.x / PAIRS;
  const bool worker = g < GROUPS;
  const int c = pair % CIN, n = pair / CIN;
  float acc[NKY][KW];
#pragma unroll
  for (int i = 0; i < NKY; ++i)
#pragma unroll
    for (int k = 0; k < KW; ++k) acc[i][k] = 0.f;
  const int rblocks = cdiv_dev(Ho, R), cblocks = cdiv_dev(Wo, WG_TW);
  const long items = (long)nimg * rblocks * cblocks;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int cb0 = (int)(it % cblocks);
    const int rb = (int)((it / cblocks) % rblocks);
    const int img = (int)(it / ((long)cblocks * rblocks));
    const int oy0 = rb * R, ox0 = cb0 * WG_TW;
    const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;
    __syncthreads();
    for (int i = threadIdx.x; i < PH * PWP; i += 256) {   // one patch pixel per thread: CIN channels, scattered by row
      const int px = i % PWP, py = i / PWP;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const float* src = x + (((long)img * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cpix;
#pragma unroll
      for (int ch = 0; ch < CIN; ++ch) xs[(py * CIN + ch) * PWP + px] = ok ? src[ch] : 0.f;
    }
    for (int i = threadIdx.x; i < R * WG_TW; i += 256) {
      const int col = i % WG_TW, r = i / WG_TW;
      const int oy = oy0 + r, ox = ox0 + col;
      const bool ok = oy < Ho && ox < Wo;
      const float4* src = reinterpret_cast<const float4*>(dz + (((long)img * Ho + (ok ? oy : 0)) * Wo + (ok ? ox : 0)) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = src[q];
        ds[(r * 16 + 4 * q + 0) * WG_TW + col] = ok ? v.x : 0.f;
        ds[(r * 16 + 4 * q + 1) * WG_TW + col] = ok ? v.y : 0.f;
        ds[(r * 16 + 4 * q + 2) * WG_TW + col] = ok ? v.z : 0.f;
        ds[(r * 16 + 4 * q + 3) * WG_TW + col] = ok ? v.w : 0.f;
      }
    }
    __syncthreads();
    if (worker) {
#pragma unroll 1
      for (int r = 0; r < R; ++r)
#pragma unroll 2
        for (int cb = 0; cb < WG_TW / 4; ++cb) {
          const float4 d4 = *reinterpret_cast<const float4*>(ds + (r * 16 + n) * WG_TW + cb * 4);
          const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < NKY; ++i) {
            const int ky = g + i * GROUPS;
            if (ky < KH) {
              const float4* xp = reinterpret_cast<const float4*>(xs + ((r * S + ky) * CIN + c) * PWP + cb * 4 * S);
              float xr[NF4 * 4];
#pragma unroll
              for (int q = 0; q < NF4; ++q) {
                const float4 v = xp[q];
                xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
              }
#pragma unroll
              for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][kx] += d[j] * xr[j * S + kx];
            }
          }
        }
    }
  }
  if (worker) {
#pragma unroll
    for (int i = 0; i < NKY; ++i) {
      const int ky = g + i * GROUPS;
      if (ky < KH) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int o = ((ky * KW + kx) * 16 + n) * CIN + c;
          scratch[(long)o * gridDim.x + blockIdx.x] = acc[i][kx];
        }
      }
    }
  }
}

// CIN = 16 on the matrix cores: dW[n][c](tap) = sum_px dz[px][n] * x[px*S + tap][c] is a 16 x 16 x (pixels) product per tap,
// v_mfma_f32_16x16x4_f32 with A = dz^T and B = x, both read from the channel-major LDS images with b128 loads (k-slot g of
// MFMA j holds pixel 4g + j of a 16-pixel group). Each wave owns whole 16-pixel groups and keeps the K*K accumulator tiles;
// its partial goes to scratch slot 4*block + wave, summed by conv_wgrad_reduce_kernel in a fixed order. The next item's
// pixels are fetched into registers while the current one is multiplied.
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
template <int S, int K, int R>
__global__ __launch_bounds__(256) void conv_wgrad16_mfma_kernel(const float* __restrict__ x, int nimg, int H, int W,
                                                                const float* __restrict__ dz, int Ho, int Wo, int pad,
                                                                float* __restrict__ scratch, int nslots) {
  constexpr int PH = (R - 1) * S + K, PWP = wg_pwpad(S, K), NF4 = wg_nf4(S, K), DSP = WG_TW + 4;
  constexpr int NX = PH * PWP * 4, NXF = (NX + 255) / 256, ND = R * WG_TW * 4, NDF = (ND + 255) / 256;
  __shared__ __attribute__((aligned(16))) float xs[PH * 16 * PWP];
  __shared__ __attribute__((aligned(16))) float ds[R * 16 * DSP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, g = lane >> 4;
  wg_f32x4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int rblocks = cdiv_dev(Ho, R), cblocks = cdiv_dev(Wo, WG_TW);
  const long items = (long)nimg * rblocks * cblocks;
  float4 fx[NXF], fd[NDF];
  auto fetch = [&](long it) {
    const int cb0 = (int)(it % cblocks), rb = (int)((it / cblocks) % rblocks), img = (int)(it / ((long)cblocks * rblocks));
    const int oy0 = rb * R, ox0 = cb0 * WG_TW, iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;
#pragma unroll
    for (int f = 0; f < NXF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pix = i >> 2, px = pix % PWP, py = pix / PWP;
      const int iy = iy0 + py, ix = ix0 + px;
      fx[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        fx[f] = *reinterpret_cast<const float4*>(x + (((long)img * H + iy) * W + ix) * 16 + 4 * q);
    }
#pragma unroll
    for (int f = 0; f < NDF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pix = i >> 2, col = pix % WG_TW, r = pix / WG_TW;
      const int oy = oy0 + r, ox = ox0 + col;
      fd[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < ND && oy < Ho && ox < Wo) fd[f] = *reinterpret_cast<const float4*>(dz + (((long)img * Ho + oy) * Wo + ox) * 16 + 4 * q);
    }
  };
  if ((long)blockIdx.x < items) fetch(blockIdx.x);
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NXF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pix = i >> 2, px = pix % PWP, py = pix / PWP;
      if (i < NX) {
        float* d = xs + (py * 16 + 4 * q) * PWP + px;
        d[0] = fx[f].x; d[PWP] = fx[f].y; d[2 * PWP] = fx[f].z; d[3 * PWP] = fx[f].w;
      }
    }
#pragma unroll
    for (int f = 0; f < NDF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pix = i >> 2, col = pix % WG_TW, r = pix / WG_TW;
      if (i < ND) {
        float* d = ds + (r * 16 + 4 * q) * DSP + col;
        d[0] = fd[f].x; d[DSP] = fd[f].y; d[2 * DSP] = fd[f].z; d[3 * DSP] = fd[f].w;
      }
    }
    __syncthreads();
    if (it + gridDim.x < items) fetch(it + gridDim.x);
#pragma unroll 1
    for (int grp = wave; grp < R * (WG_TW / 16); grp += 4) {
      const int r = grp / (WG_TW / 16), c0 = (grp % (WG_TW / 16)) * 16;
      const float4 d4 = *reinterpret_cast<const float4*>(ds + (r * 16 + idx) * DSP + c0 + 4 * g);
      const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const float4* xp = reinterpret_cast<const float4*>(xs + ((r * S + ky) * 16 + idx) * PWP + (c0 + 4 * g) * S);
        float xr[NF4 * 4];
#pragma unroll
        for (int q = 0; q < NF4; ++q) {
          const float4 v = xp[q];
          xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[ky * K + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j], xr[j * S + kx], acc[ky * K + kx], 0, 0, 0);
      }
    }
  }
  // D: row = output channel n = 4g + e, column = input channel c = lane & 15
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = (t * 16 + 4 * g + e) * 16 + idx;
      scratch[(long)o * nslots + blockIdx.x * 4 + wave] = acc[t][e];
    }
}

// Weight gradient of the stem on the auxiliary input (xn0, xn1, 1): 7x7, stride 2, pad 3, NHWC4. Two column sets per kernel
// row: set 1 = (kx, c in {0,1}) pairs, which are 16 consecutive floats of an interleaved patch row starting at 4*px; set 2 =
// the constant channel (a validity mask after zero padding), 8 consecutive floats starting at 2*px. Same pixel-to-k-slot
// assignment and scratch layout as conv_wgrad16_mfma_kernel.
constexpr int SW_R = 4, SW_PH = (SW_R - 1) * 2 + 7, SW_PWC = (WG_TW - 1) * 2 + 8, SW_XP = 144, SW_MP = 80, SW_DSP = WG_TW + 4;
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x4, int nimg, int H, int W,
                                                              const float* __restrict__ dz, int Ho, int Wo,
                                                              float* __restrict__ scratch, int nslots) {
  __shared__ __attribute__((aligned(16))) float xs[SW_PH * SW_XP];
  __shared__ __attribute__((aligned(16))) float ms[SW_PH * SW_MP];
  __shared__ __attribute__((aligned(16))) float ds[SW_R * 16 * SW_DSP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, g = lane >> 4;
  wg_f32x4 acc1[7], acc2[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) { acc1[t] = wg_f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t] = wg_f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int i = tid; i < SW_PH * SW_XP; i += 256) xs[i] = 0.f;   // padding columns stay zero
  for (int i = tid; i < SW_PH * SW_MP; i += 256) ms[i] = 0.f;
  const int rblocks = cdiv_dev(Ho, SW_R), cblocks = cdiv_dev(Wo, WG_TW);
  const long items = (long)nimg * rblocks * cblocks;
  constexpr int NX = SW_PH * SW_PWC, NXF = (NX + 255) / 256, ND = SW_R * WG_TW * 4, NDF = (ND + 255) / 256;
  float4 fx[NXF], fd[NDF];
  auto fetch = [&](long it) {
    const int cb0 = (int)(it % cblocks), rb = (int)((it / cblocks) % rblocks), img = (int)(it / ((long)cblocks * rblocks));
    const int oy0 = rb * SW_R, ox0 = cb0 * WG_TW, iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
#pragma unroll
    for (int f = 0; f < NXF; ++f) {
      const int i = tid + 256 * f, px = i % SW_PWC, py = i / SW_PWC;
      const int iy = iy0 + py, ix = ix0 + px;
      fx[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        fx[f] = *reinterpret_cast<const float4*>(x4 + (((long)img * H + iy) * W + ix) * 4);
    }
#pragma unroll
    for (int f = 0; f < NDF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pix = i >> 2, col = pix % WG_TW, r = pix / WG_TW;
      const int oy = oy0 + r, ox = ox0 + col;
      fd[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < ND && oy < Ho && ox < Wo) fd[f] = *reinterpret_cast<const float4*>(dz + (((long)img * Ho + oy) * Wo + ox) * 16 + 4 * q);
    }
  };
  if ((long)blockIdx.x < items) fetch(blockIdx.x);
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NXF; ++f) {
      const int i = tid + 256 * f, px = i % SW_PWC, py = i / SW_PWC;
      if (i < NX) {
        *reinterpret_cast<float2*>(xs + py * SW_XP + 2 * px) = make_float2(fx[f].x, fx[f].y);
        ms[py * SW_MP + px] = fx[f].z;
      }
    }
#pragma unroll
    for (int f = 0; f < NDF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pix = i >> 2, col = pix % WG_TW, r = pix / WG_TW;
      if (i < ND) {
        float* d = ds + (r * 16 + 4 * q) * SW_DSP + col;
        d[0] = fd[f].x; d[SW_DSP] = fd[f].y; d[2 * SW_DSP] = fd[f].z; d[3 * SW_DSP] = fd[f].w;
      }
    }
    __syncthreads();
    if (it + gridDim.x < items) fetch(it + gridDim.x);
#pragma unroll 1
    for (int grp = wave; grp < SW_R * (WG_TW / 16); grp += 4) {
      const int r = grp / (WG_TW / 16), c0 = (grp % (WG_TW / 16)) * 16;
      const float4 d4 = *reinterpret_cast<const float4*>(ds + (r * 16 + idx) * SW_DSP + c0 + 4 * g);
      const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const float* xr = xs + (r * 2 + ky) * SW_XP + 4 * (c0 + 4 * g) + idx;   // pixel 4g + j: + 4j
        const float* mr = ms + (r * 2 + ky) * SW_MP + 2 * (c0 + 4 * g) + idx;   //              + 2j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc1[ky] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j], xr[4 * j], acc1[ky], 0, 0, 0);
          acc2[ky] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j], mr[2 * j], acc2[ky], 0, 0, 0);
        }
      }
    }
  }
  // D: row n = 4g + e; set 1 column idx = 2*kx + c, set 2 column idx = kx
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 4 * g + e, slot = blockIdx.x * 4 + wave;
      if ((idx >> 1) < 7) scratch[(long)(((ky * 7 + (idx >> 1)) * 16 + n) * 3 + (idx & 1)) * nslots + slot] = acc1[ky][e];
      if (idx < 7) scratch[(long)(((ky * 7 + idx) * 16 + n) * 3 + 2) * nslots + slot] = acc2[ky][e];
    }
}

// one wave per output: sum its partials, add into dW (OIHW)
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

template <int S, int KH, int KW, int CIN, int R>
static void wgrad_launch(const float* x, int Cpix, int nimg, int H, int W, const float* dz, int Ho, int Wo, int pad,
                         float* scratch, float* dW, hipStream_t st) {
  const long items = (long)nimg * cdiv(Ho, R) * cdiv(Wo, WG_TW);
  const int nblk = (int)std::min<long>(items, WG_BLOCKS);
  hipLaunchKernelGGL((conv_wgrad_kernel<S, KH, KW, CIN, R>), dim3(nblk), dim3(256), 0, st, x, Cpix, nimg, H, W, dz, Ho, Wo, pad,
                     scratch);
  ATDN_HIP(hipGetLastError());
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(KH * KW * 16 * CIN, 4)), dim3(256), 0, st, scratch, nblk, CIN, KH, KW, dW);
  ATDN_HIP(hipGetLastError());
}

template <int S, int K, int R>
static void wgrad16_mfma_launch(const float* x, int nimg, int H, int W, const float* dz, int Ho, int Wo, int pad, float* scratch,
                                float* dW, hipStream_t st) {
  const long items = (long)nimg * cdiv(Ho, R) * cdiv(Wo, WG_TW);
  const int nblk = (int)std::min<long>(items, WG_MFMA_BLOCKS);
  hipLaunchKernelGGL((conv_wgrad16_mfma_kernel<S, K, R>), dim3(nblk), dim3(256), 0, st, x, nimg, H, W, dz, Ho, Wo, pad, scratch,
                     nblk * 4);
  ATDN_HIP(hipGetLastError());
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(K * K * 16 * 16, 4)), dim3(256), 0, st, scratch, nblk * 4, 16, K, K, dW);
  ATDN_HIP(hipGetLastError());
}

void launch_conv_wgrad(const float* x, int Cpix, int Cin, int nimg, int H, int W, const float* dz, int Ho, int Wo, int KH,
                       int KW, int stride, int pad, float* scratch, float* dW, hipStream_t st) {
  ATDN_CHECK(Cpix >= Cin && Cpix % 4 == 0, "conv_wgrad: channel layout");
  static const bool mfma = !(getenv("ATDN_TRAIN_WGRAD_MFMA") && getenv("ATDN_TRAIN_WGRAD_MFMA")[0] == '0');
  if (mfma && Cin == 16 && Cpix == 16) {
    if (KH == 3 && KW == 3 && stride == 1) return wgrad16_mfma_launch<1, 3, 8>(x, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
    if (KH == 3 && KW == 3 && stride == 2) return wgrad16_mfma_launch<2, 3, 4>(x, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
    if (KH == 1 && KW == 1 && stride == 2) return wgrad16_mfma_launch<2, 1, 4>(x, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
  }
  if (mfma && Cin == 3 && Cpix == 4 && KH == 7 && KW == 7 && stride == 2 && pad == 3) {
    const long items = (long)nimg * cdiv(Ho, SW_R) * cdiv(Wo, WG_TW);
    const int nblk = (int)std::min<long>(items, WG_MFMA_BLOCKS);
    hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3(nblk), dim3(256), 0, st, x, nimg, H, W, dz, Ho, Wo, scratch, nblk * 4);
    ATDN_HIP(hipGetLastError());
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(49 * 16 * 3, 4)), dim3(256), 0, st, scratch, nblk * 4, 3, 7, 7, dW);
    ATDN_HIP(hipGetLastError());
    return;
  }
  if (Cin == 16 && KH == 3 && KW == 3 && stride == 1) wgrad_launch<1, 3, 3, 16, 4>(x, Cpix, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
  else if (Cin == 16 && KH == 3 && KW == 3 && stride == 2) wgrad_launch<2, 3, 3, 16, 4>(x, Cpix, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
  else if (Cin == 16 && KH == 3 && KW == 3 && stride == 3) wgrad_launch<3, 3, 3, 16, 2>(x, Cpix, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
  else if (Cin == 16 && KH == 1 && KW == 1 && stride == 2) wgrad_launch<2, 1, 1, 16, 4>(x, Cpix, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
  else if (Cin == 3 && KH == 7 && KW == 7 && stride == 2) wgrad_launch<2, 7, 7, 3, 4>(x, Cpix, nimg, H, W, dz, Ho, Wo, pad, scratch, dW, st);
  else throw Error("conv_wgrad: no kernel for this convolution shape");
}

// The stem sees x0[c] = xn[c]*w[c] + b[c] (depthwise 1x1 on the normalised flow, zero padding applied AFTER it).
// With A = weight gradient of the stem conv taken on the auxiliary input (xn0, xn1, 1): A[n][c][tap], c = 0..2,
//   dW1[n][c][tap] += w[c]*A[n][c][tap] + b[c]*A[n][2][tap]
//   dw[c] += sum_{n,tap} W1[n][c][tap]*A[n][c][tap],   db[c] += sum_{n,tap} W1[n][c][tap]*A[n][2][tap]
// which is what back-propagating through the transposed stem convolution would give, without computing dx0.
__global__ __launch_bounds__(256) void stem_combine_kernel(const float* __restrict__ A, const float* __restrict__ W1,
                                                           const float* __restrict__ w, const float* __restrict__ b, int taps,
                                                           float* __restrict__ dW1, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[256][4];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 16 * 2 * taps; i += 256) {
    const int tap = i % taps, c = (i / taps) & 1, n = i / (2 * taps);
    const float a = A[((long)n * 3 + c) * taps + tap], ones = A[((long)n * 3 + 2) * taps + tap];
    const float wv = W1[i];
    dW1[i] += w[c] * a + b[c] * ones;
    s[c] += wv * a;
    s[2 + c] += wv * ones;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (threadIdx.x < 4) {
    float acc = 0.f;
    for (int t = 0; t < 256; ++t) acc += red[t][threadIdx.x];
    if (threadIdx.x < 2) dw[threadIdx.x] += acc; else db[threadIdx.x - 2] += acc;
  }
}
void launch_stem_combine(const float* A, const float* W1, const float* w, const float* b, int taps, float* dW1, float* dw,
                         float* db, hipStream_t st) {
  hipLaunchKernelGGL(stem_combine_kernel, dim3(1), dim3(256), 0, st, A, W1, w, b, taps, dW1, dw, db);
  ATDN_HIP(hipGetLastError());
}

// (xn0, xn1, 1, 0) per pixel: the auxiliary stem input for the weight gradients
__global__ void prep_flow_aux_kernel(const float* __restrict__ flow, int B, long HW, float sx, float sy, float4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  const long img = i / HW, p = i - img * HW;
  const float* s = flow + img * 2 * HW + p;
  out[i] = make_float4(s[0] / sx, s[HW] / sy, 1.f, 0.f);
}
void launch_prep_flow_aux(const float* flow, int nimg, int H, int W, float* out4, hipStream_t st) {
  const long n = (long)nimg * H * W;
  hipLaunchKernelGGL(prep_flow_aux_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, flow, nimg, (long)H * W, 58.1837f,
                     17.7647f, reinterpret_cast<float4*>(out4));
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
// A not transposed and K a multiple of 8: v_mfma_f32_32x32x2_f32, one 32x32 tile of C per block, K split over the four
// waves and summed through LDS in a fixed order. The recurrent products of the LSTMs (M = batch, K up to 2048) would
// otherwise run on a handful of blocks. K order inside a step of 8: MFMA j holds k = kb + 4h + j in slot h = lane >> 5, so
// every lane feeds four MFMAs from one float4 of its row.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void gemm_rows_kernel(int transB, int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                        float beta, const float* __restrict__ bias) {
  __shared__ float red[4][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int steps = K / 8, per = (steps + 3) / 4;
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  const bool mok = m0 + r < M, nok = n0 + r < N;
  const float* arow = A + (long)(mok ? m0 + r : 0) * lda + 4 * h;
  const float* brow = transB ? B + (long)(nok ? n0 + r : 0) * ldb + 4 * h : B + (long)(4 * h) * ldb + (nok ? n0 + r : 0);
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int sidx = s0; sidx < s1; ++sidx) {
    const int kb = sidx * 8;
    float4 a = *reinterpret_cast<const float4*>(arow + kb);
    float4 b;
    if (transB) b = *reinterpret_cast<const float4*>(brow + kb);
    else { const float* q = brow + (long)kb * ldb; b = make_float4(q[0], q[ldb], q[2L * ldb], q[3L * ldb]); }
    if (!mok) a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!nok) b = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][(i >> 2) * 8 + h * 4 + (i & 3)][r] = acc[i];   // D: row, column = lane & 31
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {
    const int mm = i >> 5, nn = i & 31, m = m0 + mm, n = n0 + nn;
    if (m < M && n < N) {
      float v = ((red[0][mm][nn] + red[1][mm][nn]) + red[2][mm][nn]) + red[3][mm][nn];
      if (bias) v += bias[n];
      if (beta != 0.f) v += beta * C[(long)m * ldc + n];
      C[(long)m * ldc + n] = v;
    }
  }
}

void launch_gemm(bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                 int ldc, float beta, const float* bias, hipStream_t st) {
  const bool aligned = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0) && (!transB || ((ldb % 4 == 0) && ((uintptr_t)B % 16 == 0)));
  if (!transA && K % 8 == 0 && K >= 64 && aligned) {
    hipLaunchKernelGGL(gemm_rows_kernel, dim3(cdiv(N, 32), cdiv(M, 32)), dim3(256), 0, st, transB ? 1 : 0, M, N, K, A, lda, B,
                       ldb, C, ldc, beta, bias);
    ATDN_HIP(hipGetLastError());
    return;
  }
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
  if (i < n) a[i] = mish_fast(z[i]);
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

// ------------------------------------------------------------------------------------------------ thin 16x16 convolution
namespace atdn {
namespace {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int c16_pitch(int S) { return S == 1 ? 16 : S == 2 ? 20 : 24; }   // floats per patch pixel: conflict-free b128 reads

// Eval-mode tail of a CLVO block fused into the store (inference head, clvo.hip): TAIL 0 = none (z = acc + bias: the
// training forward, whose BatchNorm needs batch statistics first), 1 = BN(Mish(.)) with the folded affine, 2 = the
// ResidualConv tail BN2(Mish(BN1(Mish(.)) + skip)). Per-channel constants live in registers (channel = lane & 15).
struct C16Consts { float sc, sh, sc2, sh2; };
template <int TAIL>
__device__ __forceinline__ C16Consts c16_consts(const Conv16Tail& t, int n) {
  C16Consts c{1.f, 0.f, 1.f, 0.f};
  if constexpr (TAIL >= 1) { c.sc = t.sc[n]; c.sh = t.sh[n]; }
  if constexpr (TAIL == 2) { c.sc2 = t.sc2[n]; c.sh2 = t.sh2[n]; }
  return c;
}
// mish(x) = x tanh(softplus(x)) = x t / (t + 2) with t = e^x (e^x + 2): one v_exp_f32 and one v_rcp_f32 instead of the
// expf / log1pf / tanhf chain of mishf_ (for x > 20 the ratio is 1 in fp32; for x -> -inf it tends to e^x with full relative
// accuracy). The inference head spent a third of its time in these tails: encoder 0.80 -> 0.50 ms per 16 pairs; features agree
// with the libm form to 1.4e-7 relative (checksum of 16 x 512 features), golden poses within their 1e-5 (round 4). Training
// (TAIL 0 + separate BatchNorm / Mish kernels) keeps mishf_.
__device__ __forceinline__ float mish_tail_(float x) {
  const float n = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.4426950408889634f);
  const float t = n * (n + 2.f);
  return x * (t * __builtin_amdgcn_rcpf(t + 2.f));
}
template <int TAIL>
__device__ __forceinline__ float c16_tail(float v, const C16Consts& c, const float* skip, long o) {
  if constexpr (TAIL == 0) return v;
  const float y = mish_tail_(v) * c.sc + c.sh;
  if constexpr (TAIL == 1) return y;
  return mish_tail_(y + skip[o]) * c.sc2 + c.sh2;
}

// Training forward (TAIL 0) with `st.part` set: the sums of Mish(z) and Mish(z)^2 per (statistics group, channel) that the
// BatchNorm behind the convolution needs (layers/conv.py:38: bn(activation(conv(x)))) are taken from the values on their way
// to memory — the separate pass that re-read z for them was 9 % of a training iteration. A lane keeps the sums of its four
// channels while the block's tiles stay in one group (group = image / st.group_imgs: the images of one time step) and the
// block writes ONE partial row per group it met: part[group][block][2][16] (zeroed by the launcher; bn_finalize adds the rows
// in double, in a fixed order: no atomics).
struct C16StatAcc {
  float s1[4], s2[4];
  int grp;
};
__device__ __forceinline__ void c16_stat_flush(C16StatAcc& a, const Conv16Stats& st, float (*sred)[2][16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { a.s1[e] += __shfl_xor(a.s1[e], m); a.s2[e] += __shfl_xor(a.s2[e], m); }
    if (n == 0) { sred[wave][0][4 * g + e] = a.s1[e]; sred[wave][1][4 * g + e] = a.s2[e]; }
    a.s1[e] = 0.f; a.s2[e] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int which = threadIdx.x >> 4, ch = threadIdx.x & 15;
    const float t = (sred[0][which][ch] + sred[1][which][ch]) + (sred[2][which][ch] + sred[3][which][ch]);
    st.part[(((long)a.grp * gridDim.x + blockIdx.x) * 2 + which) * 16 + ch] = t;
  }
  __syncthreads();
}

// K operand order: MFMA (tap, j) holds channel 4g + j in k-slot g = lane >> 4, so a lane's float4 (channels 4g..4g+3 of
// its pixel) feeds the four MFMAs of a tap component by component.
template <int K, int S, int TH, int TW, int TAIL = 0>
__global__ __launch_bounds__(256) void conv16_kernel(const float* __restrict__ x, int nimg, int H, int W,
                                                     const float* __restrict__ w, int transposed,
                                                     const float* __restrict__ bias, int pad, int Ho, int Wo,
                                                     float* __restrict__ z, int tiles_x, int tiles_img, int ntiles,
                                                     int accumulate, const Conv16Tail tail, const Conv16Stats stat) {
  constexpr int PH = (TH - 1) * S + K, PW = (TW - 1) * S + K, PP = c16_pitch(S);
  __shared__ __attribute__((aligned(16))) float patch[PH * PW * PP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // weights into operand registers once per (persistent) block: breg[tap][j] = w(n = lane & 15, c = 4*(lane >> 4) + j, tap)
  const int n = lane & 15, g = lane >> 4;
  float breg[K * K][4];
#pragma unroll
  for (int tap = 0; tap < K * K; ++tap)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * g + j;
      breg[tap][j] = transposed ? w[((long)c * 16 + n) * K * K + (K * K - 1 - tap)] : w[((long)n * 16 + c) * K * K + tap];
    }
  // Round 5: the weights are the ROW operand and the patch the COLUMN operand, so D comes out as (row = output channel 4g + e,
  // column = pixel lane & 15): a lane holds four consecutive channels of ONE pixel and stores 16 bytes, a wave one contiguous KiB
  // (the other way round every store instruction wrote four 64-byte pieces, 4 bytes per lane). Same products, same sums.
  float bv[4];
  C16Consts cc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { bv[e] = bias ? bias[4 * g + e] : 0.f; cc[e] = c16_consts<TAIL>(tail, 4 * g + e); }
  constexpr int TILES = TH * TW / 16, TPR = TW / 16;   // 16-pixel MFMA tiles of the block tile; per row
  static_assert(TILES % 2 == 0, "two tiles per wave and trip");
  // the patch of the next block tile is fetched into registers while this one is computed
  constexpr int NV = PH * PW * 4, NF = (NV + 255) / 256;
  float4 pre[NF];
  auto fetch = [&](int bt) {
    const int img = bt / tiles_img, tloc = bt - img * tiles_img;
    const int iy0 = (tloc / tiles_x) * TH * S - pad, ix0 = (tloc % tiles_x) * TW * S - pad;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = tid + 256 * f;
      const int q = i & 3, px = (i >> 2) % PW, py = (i >> 2) / PW;
      const int iy = iy0 + py, ix = ix0 + px;
      pre[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        pre[f] = *reinterpret_cast<const float4*>(x + (((long)img * H + iy) * W + ix) * 16 + 4 * q);
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  __shared__ float sred[4][2][16];
  C16StatAcc sa{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, -1};
  const bool stats = TAIL == 0 && stat.part != nullptr;
  for (int bt = blockIdx.x; bt < ntiles; bt += gridDim.x) {
    const int img = bt / tiles_img, tloc = bt - img * tiles_img;
    const int oy0 = (tloc / tiles_x) * TH, ox0 = (tloc % tiles_x) * TW;
    if (stats) {   // (uniform over the block: every thread walks the same tiles)
      const int grp = img / stat.group_imgs;
      if (grp != sa.grp) {
        if (sa.grp >= 0) c16_stat_flush(sa, stat, sred);
        sa.grp = grp;
      }
    }
    __syncthreads();   // everyone is done with the previous patch
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = tid + 256 * f;
      if (i < NV) *reinterpret_cast<float4*>(patch + ((i >> 2) / PW * PW + (i >> 2) % PW) * PP + 4 * (i & 3)) = pre[f];
    }
    __syncthreads();
    if (bt + (int)gridDim.x < ntiles) fetch(bt + gridDim.x);
    for (int t = 2 * wave; t < TILES; t += 8) {   // two independent accumulation chains per wave
      const int ty0 = t / TPR, tx0 = (t % TPR) * 16, ty1 = (t + 1) / TPR, tx1 = ((t + 1) % TPR) * 16;
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      // in the patch operand lane & 15 is the pixel of the row
      const float* b0 = patch + ((ty0 * S) * PW + (tx0 + n) * S) * PP + 4 * g;
      const float* b1 = patch + ((ty1 * S) * PW + (tx1 + n) * S) * PP + 4 * g;
#pragma unroll
      for (int tap = 0; tap < K * K; ++tap) {
        const int off = ((tap / K) * PW + (tap % K)) * PP;
        const float4 a0 = *reinterpret_cast<const float4*>(b0 + off);
        const float4 a1 = *reinterpret_cast<const float4*>(b1 + off);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][0], a0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][0], a1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][1], a0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][1], a1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][2], a0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][2], a1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][3], a0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][3], a1.w, acc1, 0, 0, 0);
      }
      // D: row (output channel) = 4*g + e, column (pixel) = lane & 15
      auto put = [&](const f32x4_t& acc, int oy, int ox) __attribute__((always_inline)) {
        if (oy >= Ho || ox >= Wo) return;
        const long o = (((long)img * Ho + oy) * Wo + ox) * 16 + 4 * g;
        float4* pz = reinterpret_cast<float4*>(z + o);
        float v[4];
        if constexpr (TAIL == 0) {
          float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
          if (accumulate) old = *pz;
          const float od[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[e] + bv[e] + od[e];
          if (stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float m = mish_fast(v[e]); sa.s1[e] += m; sa.s2[e] += m * m; }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = c16_tail<TAIL>(acc[e] + bv[e], cc[e], tail.skip, o + e);
        }
        *pz = make_float4(v[0], v[1], v[2], v[3]);
      };
      put(acc0, oy0 + ty0, ox0 + tx0 + n);
      put(acc1, oy0 + ty1, ox0 + tx1 + n);
    }
  }
  if (stats && sa.grp >= 0) c16_stat_flush(sa, stat, sred);
}

template <int K, int S, int TH, int TW, int TAIL = 0>
void conv16_launch(const float* x, int nimg, int H, int W, const float* w, bool transposed, const float* bias, int pad,
                   float* z, bool accumulate, hipStream_t st, const Conv16Tail& tail = Conv16Tail{}, Conv16Stats* stat = nullptr) {
  const int Ho = (H + 2 * pad - K) / S + 1, Wo = (W + 2 * pad - K) / S + 1;
  const int tx = cdiv(Wo, TW), ty = cdiv(Ho, TH);
  const int ntiles = nimg * tx * ty;
  const int grid = ntiles < 256 * 3 ? ntiles : 256 * 3;   // persistent blocks: the weights are loaded into registers once
  Conv16Stats sv;
  if (stat && stat->part) {
    ATDN_CHECK(TAIL == 0 && !accumulate && stat->group_imgs >= 1 && nimg % stat->group_imgs == 0, "conv16 statistics: plain training forward only");
    const int groups = nimg / stat->group_imgs;
    ATDN_CHECK((long)groups * grid * 32 <= stat->capacity, "conv16 statistics: partial buffer too small");
    ATDN_HIP(hipMemsetAsync(stat->part, 0, (size_t)groups * grid * 32 * sizeof(float), st));   // blocks write the groups they meet
    stat->rows = grid;
    sv = *stat;
  }
  hipLaunchKernelGGL((conv16_kernel<K, S, TH, TW, TAIL>), dim3(grid), dim3(256), 0, st, x, nimg, H, W, w, transposed ? 1 : 0, bias,
                     pad, Ho, Wo, z, tx, tx * ty, ntiles, accumulate ? 1 : 0, tail, sv);
  ATDN_HIP(hipGetLastError());
}
}  // namespace

// Stem of the CLVO encoder: 7x7, stride 2, pad 3, 2 -> 16 channels on NHWC4 input (channels 2, 3 unused). A patch row in
// LDS holds (column, channel) pairs back to back, so the 14 (kx, c) products of one kernel row of one output pixel are 14
// consecutive floats starting at 4*px: four MFMAs per kernel row (k-slot g of MFMA j = pair index 4g + j, the last two
// pairs carry zero weights) fed by one ds_read_b128.
namespace {
constexpr int ST_TH = 8, ST_TW = 64, ST_PH = (ST_TH - 1) * 2 + 7, ST_PWC = (ST_TW - 1) * 2 + 8, ST_ROWP = ST_PWC * 2;
template <int TAIL>
__global__ __launch_bounds__(256) void stem16_kernel(const float* __restrict__ x, int nimg, int H, int W,
                                                     const float* __restrict__ w /*[16][2][7][7]*/,
                                                     const float* __restrict__ bias, int Ho, int Wo, float* __restrict__ z,
                                                     int tiles_x, int tiles_img, int ntiles, const Conv16Tail tail,
                                                     const Conv16Stats stat) {
  __shared__ __attribute__((aligned(16))) float patch[ST_PH * ST_ROWP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
  float breg[7][4];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 4 * g + j, kx = f >> 1, c = f & 1;
      breg[ky][j] = kx < 7 ? w[(((long)n * 2 + c) * 7 + ky) * 7 + kx] : 0.f;
    }
  float bv[4];   // (weights as the row operand: a lane ends up with channels 4g..4g+3 of one pixel, see conv16_kernel)
  C16Consts cc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { bv[e] = bias ? bias[4 * g + e] : 0.f; cc[e] = c16_consts<TAIL>(tail, 4 * g + e); }
  constexpr int NV = ST_PH * ST_PWC, NF = (NV + 255) / 256;
  float2 pre[NF];
  auto fetch = [&](int bt) {
    const int img = bt / tiles_img, tloc = bt - img * tiles_img;
    const int iy0 = (tloc / tiles_x) * ST_TH * 2 - 3, ix0 = (tloc % tiles_x) * ST_TW * 2 - 3;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = tid + 256 * f, px = i % ST_PWC, py = i / ST_PWC;
      const int iy = iy0 + py, ix = ix0 + px;
      pre[f] = make_float2(0.f, 0.f);
      if (i < NV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        pre[f] = *reinterpret_cast<const float2*>(x + (((long)img * H + iy) * W + ix) * 4);
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  constexpr int TILES = ST_TH * ST_TW / 16, TPR = ST_TW / 16;
  __shared__ float sred[4][2][16];
  C16StatAcc sa{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, -1};
  const bool stats = TAIL == 0 && stat.part != nullptr;
  for (int bt = blockIdx.x; bt < ntiles; bt += gridDim.x) {
    const int img = bt / tiles_img, tloc = bt - img * tiles_img;
    const int oy0 = (tloc / tiles_x) * ST_TH, ox0 = (tloc % tiles_x) * ST_TW;
    if (stats) {
      const int grp = img / stat.group_imgs;
      if (grp != sa.grp) {
        if (sa.grp >= 0) c16_stat_flush(sa, stat, sred);
        sa.grp = grp;
      }
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = tid + 256 * f;
      if (i < NV) *reinterpret_cast<float2*>(patch + (i / ST_PWC) * ST_ROWP + (i % ST_PWC) * 2) = pre[f];
    }
    __syncthreads();
    if (bt + (int)gridDim.x < ntiles) fetch(bt + gridDim.x);
    for (int t = 2 * wave; t < TILES; t += 8) {
      const int ty0 = t / TPR, tx0 = (t % TPR) * 16, ty1 = (t + 1) / TPR, tx1 = ((t + 1) % TPR) * 16;
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* b0 = patch + (ty0 * 2) * ST_ROWP + 4 * (tx0 + n) + 4 * g;   // lane & 15 = pixel of the patch operand
      const float* b1 = patch + (ty1 * 2) * ST_ROWP + 4 * (tx1 + n) + 4 * g;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const float4 a0 = *reinterpret_cast<const float4*>(b0 + ky * ST_ROWP);
        const float4 a1 = *reinterpret_cast<const float4*>(b1 + ky * ST_ROWP);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][0], a0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][0], a1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][1], a0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][1], a1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][2], a0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][2], a1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][3], a0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[ky][3], a1.w, acc1, 0, 0, 0);
      }
      auto put = [&](const f32x4_t& acc, int oy, int ox) __attribute__((always_inline)) {
        if (oy >= Ho || ox >= Wo) return;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c16_tail<TAIL>(acc[e] + bv[e], cc[e], nullptr, 0);
        *reinterpret_cast<float4*>(z + (((long)img * Ho + oy) * Wo + ox) * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        if (stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float m = mish_fast(v[e]); sa.s1[e] += m; sa.s2[e] += m * m; }
        }
      };
      put(acc0, oy0 + ty0, ox0 + tx0 + n);
      put(acc1, oy0 + ty1, ox0 + tx1 + n);
    }
  }
  if (stats && sa.grp >= 0) c16_stat_flush(sa, stat, sred);
}
}  // namespace

void launch_stem16(const float* x4, int nimg, int H, int W, const float* w, const float* bias, float* z, hipStream_t st,
                   const Conv16Tail* tail, Conv16Stats* stat) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tx = cdiv(Wo, ST_TW), ty = cdiv(Ho, ST_TH), ntiles = nimg * tx * ty;
  const int grid = ntiles < 256 * 4 ? ntiles : 256 * 4;
  if (tail) {
    ATDN_CHECK(tail->sc && tail->sh && !tail->skip, "stem tail is BN(Mish(.))");
    ATDN_CHECK(!stat || !stat->part, "stem statistics: training forward only");
    hipLaunchKernelGGL(stem16_kernel<1>, dim3(grid), dim3(256), 0, st, x4, nimg, H, W, w, bias, Ho, Wo, z, tx, tx * ty, ntiles, *tail,
                       Conv16Stats{});
  } else {
    Conv16Stats sv;
    if (stat && stat->part) {
      ATDN_CHECK(stat->group_imgs >= 1 && nimg % stat->group_imgs == 0, "stem statistics: whole groups of images");
      const int groups = nimg / stat->group_imgs;
      ATDN_CHECK((long)groups * grid * 32 <= stat->capacity, "stem statistics: partial buffer too small");
      ATDN_HIP(hipMemsetAsync(stat->part, 0, (size_t)groups * grid * 32 * sizeof(float), st));
      stat->rows = grid;
      sv = *stat;
    }
    hipLaunchKernelGGL(stem16_kernel<0>, dim3(grid), dim3(256), 0, st, x4, nimg, H, W, w, bias, Ho, Wo, z, tx, tx * ty, ntiles,
                       Conv16Tail{}, sv);
  }
  ATDN_HIP(hipGetLastError());
}

// Data gradient of a stride-2 16 -> 16 convolution without the zero-stuffed map: dx[y][x][c] = sum over the taps whose
// source (y + PAD - ky)/2, (x + PAD - kx)/2 is integral. Output pixels of one row and one column parity share their tap
// list (1, 2, 2 or 4 taps for 3x3), and 16 of them read 16 consecutive dz columns, so each parity class is a small
// stride-1 convolution on the dz patch. Wave w owns rows 2w and 2w+1 of the 8 x 64 tile (every class once).
namespace {
template <int K, int PAD>
__global__ __launch_bounds__(256) void tconv16_s2_kernel(const float* __restrict__ dz, int nimg, int Ho, int Wo,
                                                         const float* __restrict__ w, int H, int W, int accumulate,
                                                         float* __restrict__ dx, int tiles_x, int tiles_img, int ntiles) {
  constexpr int TH = 8, TW = 64, PH = TH / 2 + 2, PW = TW / 2 + 2;
  __shared__ __attribute__((aligned(16))) float patch[PH * PW * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, g = lane >> 4;
  float breg[K * K][4];   // contraction over n: slot g of MFMA j = output channel 4g + j; row = input channel idx
#pragma unroll
  for (int tap = 0; tap < K * K; ++tap)
#pragma unroll
    for (int j = 0; j < 4; ++j) breg[tap][j] = w[((long)(4 * g + j) * 16 + idx) * K * K + tap];
  constexpr int NV = PH * PW * 4, NF = (NV + 255) / 256;
  float4 pre[NF];
  auto fetch = [&](int bt) {
    const int img = bt / tiles_img, tloc = bt - img * tiles_img;
    const int oyb = (tloc / tiles_x) * (TH / 2) - 1, oxb = (tloc % tiles_x) * (TW / 2) - 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = tid + 256 * f, q = i & 3, pc = (i >> 2) % PW, pr = (i >> 2) / PW;
      const int oy = oyb + pr, ox = oxb + pc;
      pre[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NV && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo)
        pre[f] = *reinterpret_cast<const float4*>(dz + (((long)img * Ho + oy) * Wo + ox) * 16 + 4 * q);
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int bt = blockIdx.x; bt < ntiles; bt += gridDim.x) {
    const int img = bt / tiles_img, tloc = bt - img * tiles_img;
    const int y0 = (tloc / tiles_x) * TH, x0 = (tloc % tiles_x) * TW;
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = tid + 256 * f;
      if (i < NV) *reinterpret_cast<float4*>(patch + (i >> 2) * 16 + 4 * (i & 3)) = pre[f];
    }
    __syncthreads();
    if (bt + (int)gridDim.x < ntiles) fetch(bt + gridDim.x);
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      const int ly = 2 * wave + py, y = y0 + ly;
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          if ((py + PAD - ky) & 1) continue;
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            if ((px + PAD - kx) & 1) continue;
            // (ly + PAD - ky)/2 and (px + PAD - kx)/2 are exact; the patch starts one dz row / column before the tile
            const float* b = patch + ((((ly + PAD - ky) >> 1) + 1) * PW + idx + ((px + PAD - kx) >> 1) + 1) * 16 + 4 * g;
            const float4 a0 = *reinterpret_cast<const float4*>(b);
            const float4 a1 = *reinterpret_cast<const float4*>(b + 16 * 16);
            const int tap = ky * K + kx;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][0], a0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][0], a1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][1], a0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][1], a1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][2], a0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][2], a1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][3], a0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tap][3], a1.w, acc1, 0, 0, 0);
          }
        }
        if (y < H) {   // (weights as the row operand: the lane holds channels 4g..4g+3 of pixel idx, see conv16_kernel)
          const int xa = x0 + 2 * idx + px, xb = xa + 32;
          float4* pa = reinterpret_cast<float4*>(dx + (((long)img * H + y) * W + xa) * 16 + 4 * g);
          auto put = [&](float4* q, const f32x4_t& acc) __attribute__((always_inline)) {
            float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accumulate) old = *q;
            *q = make_float4(acc[0] + old.x, acc[1] + old.y, acc[2] + old.z, acc[3] + old.w);
          };
          if (xa < W) put(pa, acc0);
          if (xb < W) put(pa + 32 * 4, acc1);
        }
      }
    }
  }
}
}  // namespace

void launch_tconv16_s2(const float* dz, int nimg, int Ho, int Wo, const float* w, int K, int pad, int H, int W, bool accumulate,
                       float* dx, hipStream_t st) {
  const int tx = cdiv(W, 64), ty = cdiv(H, 8), ntiles = nimg * tx * ty;
  const int grid = ntiles < 256 * 4 ? ntiles : 256 * 4;
  if (K == 3 && pad == 1)
    hipLaunchKernelGGL((tconv16_s2_kernel<3, 1>), dim3(grid), dim3(256), 0, st, dz, nimg, Ho, Wo, w, H, W, accumulate ? 1 : 0, dx,
                       tx, tx * ty, ntiles);
  else if (K == 1 && pad == 0)
    hipLaunchKernelGGL((tconv16_s2_kernel<1, 0>), dim3(grid), dim3(256), 0, st, dz, nimg, Ho, Wo, w, H, W, accumulate ? 1 : 0, dx,
                       tx, tx * ty, ntiles);
  else throw Error("tconv16_s2: no kernel for this shape");
  ATDN_HIP(hipGetLastError());
}

void launch_conv16(const float* x, int nimg, int H, int W, const float* w, bool transposed, const float* bias, int K, int S,
                   int pad, float* z, hipStream_t st, bool accumulate, Conv16Stats* stat) {
  const Conv16Tail nt{};
  if (K == 3 && S == 1) conv16_launch<3, 1, 8, 64>(x, nimg, H, W, w, transposed, bias, pad, z, accumulate, st, nt, stat);
  else if (K == 3 && S == 2) conv16_launch<3, 2, 4, 32>(x, nimg, H, W, w, transposed, bias, pad, z, accumulate, st, nt, stat);
  else if (K == 3 && S == 3) conv16_launch<3, 3, 2, 32>(x, nimg, H, W, w, transposed, bias, pad, z, accumulate, st, nt, stat);
  else if (K == 1 && S == 2) conv16_launch<1, 2, 4, 32>(x, nimg, H, W, w, transposed, bias, pad, z, accumulate, st, nt, stat);
  else if (K == 1 && S == 1) conv16_launch<1, 1, 8, 64>(x, nimg, H, W, w, transposed, bias, pad, z, accumulate, st, nt, stat);
  else throw Error("conv16: no kernel for this shape");
}

void launch_conv16_eval(const float* x, int nimg, int H, int W, const float* w, const float* bias, int K, int S, int pad,
                        const Conv16Tail& tail, float* z, hipStream_t st) {
  ATDN_CHECK(tail.sc && tail.sh, "eval tail needs the folded BatchNorm affine");
  const bool res = tail.skip != nullptr;
  ATDN_CHECK(!res || (tail.sc2 && tail.sh2), "residual tail needs the second affine");
  if (K == 3 && S == 1 && !res) conv16_launch<3, 1, 8, 64, 1>(x, nimg, H, W, w, false, bias, pad, z, false, st, tail);
  else if (K == 3 && S == 2 && res) conv16_launch<3, 2, 4, 32, 2>(x, nimg, H, W, w, false, bias, pad, z, false, st, tail);
  else if (K == 3 && S == 3 && !res) conv16_launch<3, 3, 2, 32, 1>(x, nimg, H, W, w, false, bias, pad, z, false, st, tail);
  else throw Error("conv16_eval: no kernel for this shape");
}

}  // namespace atdn
