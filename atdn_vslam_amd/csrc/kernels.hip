// Non-GEMM kernels of the odometry path, gfx950. Wave = 64 lanes throughout.
#include "kernels.h"
#include "sf.h"

namespace atdn {

bool& sf_fast_mode() {
  static thread_local bool fast = false;
  return fast;
}

// ------------------------------------------------------------------ block reductions (deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* sm) {  // blockDim 256
  v = MAX ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const float a = sm[0], b = sm[1], c = sm[2], d = sm[3];
  return MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : ((a + b) + (c + d));
}

// ------------------------------------------------------------------ frame preparation
__global__ void prep_images_kernel(const float* __restrict__ im1, const float* __restrict__ im2, int B, int n2, long HW,
                                   float4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)(B + n2) * HW) return;
  const long img = i / HW, p = i - img * HW;
  const float* src = (img < B ? im1 + img * 3 * HW : im2 + (img - B) * 3 * HW) + p;
  float4 v;
  v.x = 2.f * (src[0] / 255.0f) - 1.0f;
  v.y = 2.f * (src[HW] / 255.0f) - 1.0f;
  v.z = 2.f * (src[2 * HW] / 255.0f) - 1.0f;
  v.w = 0.f;
  out[i] = v;
}
void launch_prep_images(const float* im1, const float* im2, int B, int H, int W, float* img4, hipStream_t st, int n2) {
  const long n = (long)(B + n2) * H * W;
  hipLaunchKernelGGL(prep_images_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, im1, im2, B, n2, (long)H * W,
                     reinterpret_cast<float4*>(img4));
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ instance norm
// mean / rstd [nimg][C] from the conv epilogues' per-(32-row group, channel) partials (sum, M2 about the group mean,
// valid rows), in two levels so that the ~3700 groups per channel are spread over the chip:
// level 1: block (channel slab of 64, image, split z of FIN_SPLIT) folds its share of the groups in fp64 — lane =
//   channel (coalesced partial reads), 16 waves stride over the groups, then merge through LDS in a fixed order — into
//   (n, S1 = sum of sums, Sq = sum of sum_g^2 / n_g, M2 = sum of M2_g);
// level 2: one thread per (image, channel) adds the FIN_SPLIT quadruples in a fixed order (deterministic) and writes
//   mean = S1 / n,   var = (M2 + Sq - S1^2 / n) / HW.
// The between-group term comes from the sum-of-squares identity: in fp64 its cancellation costs mean^2 / var * 1e-16,
// and it needs no division per group. The pass is latency-bound (a thread walks its groups one dependent load pair at a
// time): 32 slabs instead of 8 took level 1 from 17 to 8 us on the largest layer set. A single launch with a
// last-block-merges counter was measured SLOWER (28 us): a device-scope release on this chip writes back the XCD's
// whole L2.
constexpr int FIN_SPLIT = 32;
__global__ __launch_bounds__(1024) void in_finalize_cnt_kernel(const float* __restrict__ ps,
                                                               const float* __restrict__ pm2,
                                                               const float* __restrict__ pc, int groups, int C,
                                                               int HW, double* __restrict__ part) {
  __shared__ double s_n[16][64], s_s1[16][64], s_sq[16][64], s_m2[16][64];
  const int img = blockIdx.y, z = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int per = (groups + FIN_SPLIT - 1) / FIN_SPLIT;
  const int g0 = z * per, g1 = min(groups, g0 + per);
  double n = 0.0, s1 = 0.0, sq = 0.0, m2 = 0.0;
  if (c < C) {
    const float* s = ps + (long)img * groups * C + c;
    const float* m = pm2 + (long)img * groups * C + c;
    const float* cn = pc ? pc + (long)img * groups : nullptr;
#pragma unroll 8
    for (int g = g0 + wv; g < g1; g += 16) {
      // valid rows of the group: from the kernel's count, or (1-D M tiling) the rows of the image inside it
      const float cnt = cn ? cn[g] : (float)max(0, min(32, HW - g * 32));
      if (cnt > 0.f) {
        const double sg = (double)s[(long)g * C];
        n += (double)cnt;
        s1 += sg;
        sq += sg * sg * (cnt == 32.f ? 0.03125 : 1.0 / (double)cnt);
        m2 += (double)m[(long)g * C];
      }
    }
  }
  s_n[wv][lane] = n; s_s1[wv][lane] = s1; s_sq[wv][lane] = sq; s_m2[wv][lane] = m2;
  __syncthreads();
  if (wv == 0 && c < C) {
    n = 0.0; s1 = 0.0; sq = 0.0; m2 = 0.0;
    for (int k = 0; k < 16; ++k) { n += s_n[k][lane]; s1 += s_s1[k][lane]; sq += s_sq[k][lane]; m2 += s_m2[k][lane]; }
    double* o = part + (((long)img * FIN_SPLIT + z) * C + c) * 4;
    o[0] = n; o[1] = s1; o[2] = sq; o[3] = m2;
  }
}
__global__ void in_finalize_merge_kernel(const double* __restrict__ part, int nimg, int HW, int C, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg * C) return;
  const int img = i / C, c = i - img * C;
  double n = 0.0, s1 = 0.0, sq = 0.0, m2 = 0.0;
  for (int z = 0; z < FIN_SPLIT; ++z) {
    const double* p = part + (((long)img * FIN_SPLIT + z) * C + c) * 4;
    n += p[0]; s1 += p[1]; sq += p[2]; m2 += p[3];
  }
  const double mu = n > 0.0 ? s1 / n : 0.0;
  const double between = n > 0.0 ? fmax(sq - s1 * s1 / n, 0.0) : 0.0;
  const float mu_f = (float)mu, rs_f = (float)(1.0 / sqrt((m2 + between) / (double)HW + (double)eps));
  mean[i] = mu_f;
  rstd[i] = rs_f;
  // a NaN / infinity anywhere in the raw (fp32, un-clamped) channel shows here: the consumers' normalise-on-load loaders and
  // in_apply_sf_kernel would turn it into zeros without an alarm of their own
  sf_report(!(fabsf(mu_f) <= 3.0e38f) | !(rs_f <= 3.0e38f));
}
void launch_in_finalize_cnt(const float* part_sum, const float* part_m2, const float* part_cnt, int nimg,
                            int groups_per_img, int HW, int C, float eps, float* mean, float* rstd, double* scratch,
                            hipStream_t st) {
  hipLaunchKernelGGL(in_finalize_cnt_kernel, dim3(cdiv(C, 64), nimg, FIN_SPLIT), dim3(1024), 0, st, part_sum, part_m2,
                     part_cnt, groups_per_img, C, HW, scratch);
  ATDN_HIP(hipGetLastError());
  hipLaunchKernelGGL(in_finalize_merge_kernel, dim3(cdiv(nimg * C, 128)), dim3(128), 0, st, scratch, nimg, HW, C, eps,
                     mean, rstd);
  ATDN_HIP(hipGetLastError());
}

__global__ void in_apply_kernel(float4* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float4* __restrict__ res, const float* __restrict__ rmean,
                                const float* __restrict__ rrstd, long per_img4, int C, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long img = i / per_img4;
    const int c = (int)((i * 4) % C);
    const float* mu = mean + img * C + c;
    const float* rs = rstd + img * C + c;
    float4 v = x[i];
    v.x = fmaxf((v.x - mu[0]) * rs[0], 0.f);
    v.y = fmaxf((v.y - mu[1]) * rs[1], 0.f);
    v.z = fmaxf((v.z - mu[2]) * rs[2], 0.f);
    v.w = fmaxf((v.w - mu[3]) * rs[3], 0.f);
    if (res) {
      float4 r = res[i];
      if (rmean) {
        const float* m2 = rmean + img * C + c;
        const float* r2 = rrstd + img * C + c;
        r.x = (r.x - m2[0]) * r2[0]; r.y = (r.y - m2[1]) * r2[1];
        r.z = (r.z - m2[2]) * r2[2]; r.w = (r.w - m2[3]) * r2[3];
      }
      v.x = fmaxf(r.x + v.x, 0.f); v.y = fmaxf(r.y + v.y, 0.f);
      v.z = fmaxf(r.z + v.z, 0.f); v.w = fmaxf(r.w + v.w, 0.f);
    }
    x[i] = v;
  }
}
void launch_in_apply(float* x, const float* mean, const float* rstd, const float* res, const float* rmean,
                     const float* rrstd, int nimg, long HW, int C, hipStream_t st) {
  ATDN_CHECK(C % 4 == 0, "in_apply needs C % 4 == 0");
  const long per_img4 = HW * C / 4, total4 = per_img4 * nimg;
  const int grid = (int)std::min<long>(cdivl(total4, 256), 256 * 16);
  hipLaunchKernelGGL(in_apply_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<float4*>(x), mean, rstd,
                     reinterpret_cast<const float4*>(res), rmean, rrstd, per_img4, C, total4);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ pyramid pooling
__global__ void avgpool_kernel(const float* __restrict__ src, int H, int W, int Ho, int Wo, float* __restrict__ dst,
                               long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (Ho * Wo);
    const int rem = (int)(i - row * (Ho * Wo));
    const int y = rem / Wo, x = rem - y * Wo;
    const float* p = src + row * ((long)H * W) + (long)(2 * y) * W + 2 * x;
    dst[i] = (((p[0] + p[1]) + p[W]) + p[W + 1]) * 0.25f;
  }
}
void launch_avgpool(const float* src, int H, int W, float* dst, long rows, hipStream_t st) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = rows * Ho * Wo;
  const int grid = (int)std::min<long>(cdivl(total, 256), 256 * 32);
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid), dim3(256), 0, st, src, H, W, Ho, Wo, dst, total);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ correlation-pyramid lookup
// One wave per (source pixel, level): a 12x12 window around the sampling centre is staged in LDS (cells
// outside the map are the zero padding), then lanes produce the 81 bilinear samples and store them as one
// contiguous run. Sample coordinates follow the reference's arithmetic step by step — centroid/2^l + delta
// (corr.py:44-46), 2x/(W-1)-1 (utils.py:63-64), then grid_sample's align_corners un-normalisation
// (x+1)*((W-1)/2) — so floor() and the four weights see the same fp32 values as the oracle; the 12-wide
// window (one cell more than the nominal 10 on each side) covers a floor() that lands one off at integers.
constexpr int LK_WIN = 12;
__global__ __launch_bounds__(256) void lookup_kernel(const PyramidLevels pyr, const float* __restrict__ coords1,
                                                     long npix, float* __restrict__ out, int ldo) {
  __shared__ float win[4][LK_WIN * LK_WIN + 16];
  const int lvl = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Hl = pyr.H[lvl], Wl = pyr.W[lvl];
  const float inv = 1.0f / (float)(1 << lvl);
  const float wm1 = (float)(Wl - 1), hm1 = (float)(Hl - 1);
  const float sfx = wm1 / 2.f, sfy = hm1 / 2.f;
  float* w = win[lvl];
  for (long p = blockIdx.x; p < npix; p += gridDim.x) {
    const float xc = coords1[p * 2 + 0] * inv, yc = coords1[p * 2 + 1] * inv;
    const bool sane = (fabsf(xc) < 1.0e6f) && (fabsf(yc) < 1.0e6f);  // also rejects NaN
    const int wx0 = sane ? (int)floorf(xc) - 5 : -(1 << 24), wy0 = sane ? (int)floorf(yc) - 5 : -(1 << 24);
    const float* src = pyr.base[lvl] + p * ((long)Hl * Wl);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int c = lane + 64 * t;
      if (c < LK_WIN * LK_WIN) {
        const int wy = c / LK_WIN, wx = c - wy * LK_WIN;
        const int y = wy0 + wy, x = wx0 + wx;
        const bool ok = ((unsigned)y < (unsigned)Hl) & ((unsigned)x < (unsigned)Wl);
        w[c] = ok ? src[(long)y * Wl + x] : 0.f;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);  // this wave's window is complete (one wave owns win[lvl])
    __builtin_amdgcn_wave_barrier();
    float* o = out + p * ldo + lvl * 81;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = lane + 64 * t;
      if (k < 81) {
        const int i = k / 9, j = k - i * 9;  // i steps x, j steps y (RAFT's transposed window)
        float v = 0.f;
        if (sane) {
          const float px = xc + (float)(i - 4), py = yc + (float)(j - 4);
          const float xg = 2.f * px / wm1 - 1.f, yg = 2.f * py / hm1 - 1.f;
          const float ix = (xg + 1.f) * sfx, iy = (yg + 1.f) * sfy;
          const float xw = floorf(ix), yn = floorf(iy);
          const float ww = ix - xw, ee = 1.f - ww, nn = iy - yn, ss = 1.f - nn;
          const int lx = min(max((int)xw - wx0, 0), LK_WIN - 2), ly = min(max((int)yn - wy0, 0), LK_WIN - 2);
          const float* q = w + ly * LK_WIN + lx;
          v = ((q[0] * (ee * ss) + q[1] * (ww * ss)) + q[LK_WIN] * (ee * nn)) + q[LK_WIN + 1] * (ww * nn);
        }
        o[k] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();  // all reads done before the next pixel overwrites the window
  }
}
void launch_lookup(const PyramidLevels& pyr, const float* coords1, long npix_total, float* out, int ldo,
                   hipStream_t st) {
  ATDN_CHECK(ldo >= 324, "lookup output row too short");
  const int grid = (int)std::min<long>(npix_total, 256 * 16);
  hipLaunchKernelGGL(lookup_kernel, dim3(grid), dim3(256), 0, st, pyr, coords1, npix_total, out, ldo);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ row softmax (attention)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int n, int ld) {
  __shared__ float sm[4];
  float* row = x + (long)blockIdx.x * ld;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, row[i]);
  mx = block_reduce<true>(mx, sm);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float e = expf(row[i] - mx);
    row[i] = e;
    s += e;
  }
  s = block_reduce<false>(s, sm);
  for (int i = threadIdx.x; i < n; i += 256) row[i] = row[i] / s;
  for (int i = n + threadIdx.x; i < ld; i += 256) row[i] = 0.f;
}
void launch_softmax_rows(float* x, long rows, int n, int ld, hipStream_t st) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, x, n, ld);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ coordinate / flow state
__global__ void init_coords_kernel(const float* __restrict__ flow_init, int B, int H8, int W8,
                                   float* __restrict__ coords1, float* __restrict__ flow4, float* __restrict__ xflow,
                                   int ldx) {
  const long N = (long)H8 * W8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const long img = i / N;
  const int m = (int)(i - img * N);
  const float x0 = (float)(m % W8), y0 = (float)(m / W8);
  float fx = 0.f, fy = 0.f;
  if (flow_init) { fx = flow_init[(img * 2 + 0) * N + m]; fy = flow_init[(img * 2 + 1) * N + m]; }
  const float cx = x0 + fx, cy = y0 + fy;
  coords1[i * 2 + 0] = cx;
  coords1[i * 2 + 1] = cy;
  const float flx = cx - x0, fly = cy - y0;
  reinterpret_cast<float4*>(flow4)[i] = make_float4(flx, fly, 0.f, 0.f);
  xflow[i * ldx + 0] = flx;
  xflow[i * ldx + 1] = fly;
}
void launch_init_coords(const float* flow_init, int B, int H8, int W8, float* coords1, float* flow4, float* xflow,
                        int ldx, hipStream_t st) {
  const long n = (long)B * H8 * W8;
  hipLaunchKernelGGL(init_coords_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, flow_init, B, H8, W8,
                     coords1, flow4, xflow, ldx);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ convex upsampling
// One wave per coarse pixel, lane = (i, j) of its 8x8 output patch; mask row is 9 x 64 contiguous floats.
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ mask, const float* __restrict__ flow4,
                                                       int B, int H8, int W8, float* __restrict__ flow_low,
                                                       float* __restrict__ flow_up) {
  const long N = (long)H8 * W8;
  const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= B * N) return;
  const int lane = threadIdx.x & 63;
  const long img = p / N;
  const int m = (int)(p - img * N);
  const int h = m / W8, w = m - h * W8;
  const float* mk = mask + p * 576 + lane;
  float e[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) { e[k] = mk[k * 64]; mx = fmaxf(mx, e[k]); }
  float s = 0.f;
#pragma unroll
  // (v_exp_f32 and ONE division per output pixel: the libm expf and nine IEEE divisions were ~225 vector instructions per lane)
  for (int k = 0; k < 9; ++k) { e[k] = __builtin_amdgcn_exp2f((e[k] - mx) * 1.4426950408889634f); s += e[k]; }
  const float rs = 1.0f / s;
  float ux = 0.f, uy = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = h + k / 3 - 1, xx = w + k % 3 - 1;
    float fx = 0.f, fy = 0.f;
    if ((unsigned)yy < (unsigned)H8 && (unsigned)xx < (unsigned)W8) {
      const float4 f = reinterpret_cast<const float4*>(flow4)[img * N + (long)yy * W8 + xx];
      fx = 8.f * f.x; fy = 8.f * f.y;
    }
    const float wk = e[k] * rs;
    ux += wk * fx;
    uy += wk * fy;
  }
  const int i = lane >> 3, j = lane & 7;
  const long Hf = 8L * H8, Wf = 8L * W8;
  const long o = (img * 2) * Hf * Wf + (long)(8 * h + i) * Wf + 8 * w + j;
  flow_up[o] = ux;
  flow_up[o + Hf * Wf] = uy;
  if (lane == 0 && flow_low) {   // (flow_low == nullptr: the per-iteration predictions of GmaNet::forward_predictions)
    const float4 f = reinterpret_cast<const float4*>(flow4)[p];
    flow_low[(img * 2 + 0) * N + m] = f.x;
    flow_low[(img * 2 + 1) * N + m] = f.y;
  }
}
void launch_upsample(const float* mask, const float* flow4, int B, int H8, int W8, float* flow_low, float* flow_up,
                     hipStream_t st) {
  const long n = (long)B * H8 * W8;
  hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)cdivl(n, 4)), dim3(256), 0, st, mask, flow4, B, H8, W8, flow_low,
                     flow_up);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ CLVO head
__global__ void prep_flow_kernel(const float* __restrict__ flow, int B, long HW, float sx, float sy, const float* dw_w,
                                 const float* dw_b, float4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  const long img = i / HW, p = i - img * HW;
  const float* s = flow + img * 2 * HW + p;
  float4 v;
  v.x = (s[0] / sx) * dw_w[0] + dw_b[0];
  v.y = (s[HW] / sy) * dw_w[1] + dw_b[1];
  v.z = 0.f; v.w = 0.f;
  out[i] = v;
}
void launch_prep_flow(const float* flow, int B, int H, int W, const float* dw_w, const float* dw_b, float* out4,
                      hipStream_t st) {
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(prep_flow_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, flow, B, (long)H * W,
                     58.1837f, 17.7647f, dw_w, dw_b, reinterpret_cast<float4*>(out4));
  ATDN_HIP(hipGetLastError());
}

// MappingVAE input: NCHW 0..255 -> NHWC4 with get_rgb_norm() applied (utils/normalizations.py:4-6: x/255, then
// (x - mean)/std with the ImageNet statistics), 4th channel 0
__global__ void prep_rgb_kernel(const float* __restrict__ img, int B, long HW, float4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  const long b = i / HW, p = i - b * HW;
  const float* s = img + b * 3 * HW + p;
  float4 v;
  v.x = ((s[0] / 255.0f) - 0.485f) / 0.229f;
  v.y = ((s[HW] / 255.0f) - 0.456f) / 0.224f;
  v.z = ((s[2 * HW] / 255.0f) - 0.406f) / 0.225f;
  v.w = 0.f;
  out[i] = v;
}
void launch_prep_rgb(const float* images, int B, int H, int W, float* out4, hipStream_t st) {
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(prep_rgb_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, images, B, (long)H * W,
                     reinterpret_cast<float4*>(out4));
  ATDN_HIP(hipGetLastError());
}

// one wave per output feature n, all batch rows; K multiples of 4, rows 16-byte aligned
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ W0, const float* __restrict__ x0, int K0,
                                                     int ldx0, const float* __restrict__ W1,
                                                     const float* __restrict__ x1, int K1, int ldx1,
                                                     const float* __restrict__ b0, const float* __restrict__ b1,
                                                     int act, float* __restrict__ y, int ldy, int N, int B) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  for (int b = 0; b < B; ++b) {
    float acc = 0.f;
    const float4* w = reinterpret_cast<const float4*>(W0 + (long)n * K0);
    const float4* x = reinterpret_cast<const float4*>(x0 + (long)b * ldx0);
    for (int k = lane; k < K0 / 4; k += 64) {
      const float4 a = w[k], v = x[k];
      acc += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
    }
    if (W1) {
      const float4* w1 = reinterpret_cast<const float4*>(W1 + (long)n * K1);
      const float4* x1v = reinterpret_cast<const float4*>(x1 + (long)b * ldx1);
      for (int k = lane; k < K1 / 4; k += 64) {
        const float4 a = w1[k], v = x1v[k];
        acc += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      float v = acc + (b0 ? b0[n] : 0.f) + (b1 ? b1[n] : 0.f);
      if (act == 1) v = mishf_(v);
      y[(long)b * ldy + n] = v;
    }
  }
}
void launch_linear(const float* W0, const float* x0, int K0, int ldx0, const float* W1, const float* x1, int K1,
                   int ldx1, const float* b0, const float* b1, int act, float* y, int ldy, int N, int B,
                   hipStream_t st) {
  ATDN_CHECK(K0 % 4 == 0 && K1 % 4 == 0 && ldx0 % 4 == 0 && ldx1 % 4 == 0, "linear: K and ld must be multiples of 4");
  hipLaunchKernelGGL(linear_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, W0, x0, K0, ldx0, W1, x1, K1, ldx1, b0, b1, act,
                     y, ldy, N, B);
  ATDN_HIP(hipGetLastError());
}

// ---- the whole recurrent tail as a three-stage software pipeline, ONE launch per time step (the scan is bound by the
// launch rate, ~5.5 us per dependent launch): launch s runs lstm1 for step s (blocks [0,G)), lstm_linear + Mish for step
// s - 1 and lstm2 — input projection included — for step s - 2; every stage reads what the previous launch wrote.
// One block per hidden unit with one wave per gate (4.6 k waves per launch: the 13 MB of weights stream from L2 in ~3 us).
__global__ __launch_bounds__(256) void lstm_pipe_kernel(const LstmPipeArgs a) {
  // blocks [0,Hd): lstm1, one block per hidden unit, wave g = gate g; [Hd, Hd + Hd/4): lstm_linear, one wave per output;
  // [Hd + Hd/4, 2Hd + Hd/4): lstm2 likewise (each wave: its gate row of W_ih2 and of W_hh2)
  __shared__ float gate[4];
  const int Hd = a.Hd, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto dot = [&](const float* __restrict__ wrow, const float* __restrict__ x) {
    float acc = 0.f;
    for (int k = lane; k < Hd / 4; k += 64) {
      const float4 w = reinterpret_cast<const float4*>(wrow)[k];
      const float4 v = reinterpret_cast<const float4*>(x)[k];
      acc += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
    }
    return wave_sum(acc);
  };
  int blk = blockIdx.x;
  if (blk < Hd) {
    if (!a.do1) return;
    const int j = blk;
    for (int b = 0; b < a.B; ++b) {
      const float acc = dot(a.Whh1 + ((long)wv * Hd + j) * Hd, a.h1_in + (long)b * Hd);
      if (lane == 0) gate[wv] = a.pre1[(long)b * 4 * Hd + wv * Hd + j] + (acc + a.bhh1[wv * Hd + j]);
      __syncthreads();
      if (threadIdx.x == 0) {
        const float cn = sigmoidf_(gate[1]) * a.c1[(long)b * Hd + j] + sigmoidf_(gate[0]) * tanhf(gate[2]);
        a.c1[(long)b * Hd + j] = cn;
        a.h1_out[(long)b * Hd + j] = sigmoidf_(gate[3]) * tanhf(cn);
      }
      __syncthreads();
    }
    return;
  }
  blk -= Hd;
  if (blk < Hd / 4) {
    if (!a.do_lin) return;
    const int j = blk * 4 + wv;
    for (int b = 0; b < a.B; ++b) {
      const float acc = dot(a.Wlin + (long)j * Hd, a.lin_in + (long)b * Hd);
      if (lane == 0) a.lin_out[(long)b * Hd + j] = mishf_(acc + a.blin[j]);
    }
    return;
  }
  blk -= Hd / 4;
  if (!a.do2) return;
  const int j = blk;
  for (int b = 0; b < a.B; ++b) {
    const float ax = dot(a.Wih2 + ((long)wv * Hd + j) * Hd, a.x2_in + (long)b * Hd);
    const float ah = dot(a.Whh2 + ((long)wv * Hd + j) * Hd, a.h2_in + (long)b * Hd);
    if (lane == 0) gate[wv] = (ax + a.bih2[wv * Hd + j]) + (ah + a.bhh2[wv * Hd + j]);
    __syncthreads();
    if (threadIdx.x == 0) {
      const float cn = sigmoidf_(gate[1]) * a.c2[(long)b * Hd + j] + sigmoidf_(gate[0]) * tanhf(gate[2]);
      a.c2[(long)b * Hd + j] = cn;
      a.h2_out[(long)b * Hd + j] = sigmoidf_(gate[3]) * tanhf(cn);
    }
    __syncthreads();
  }
}
void launch_lstm_pipe(const LstmPipeArgs& a, hipStream_t st) {
  ATDN_CHECK(a.Hd % 16 == 0 && a.B >= 1, "lstm_pipe: bad arguments");
  hipLaunchKernelGGL(lstm_pipe_kernel, dim3(2 * a.Hd + a.Hd / 4), dim3(256), 0, st, a);
  ATDN_HIP(hipGetLastError());
}

// block = one batch row; waves 0-1 run the rotation head, waves 2-3 the translation head
__global__ __launch_bounds__(256) void mlp_heads_kernel(const float* __restrict__ h2, MlpHead rot, MlpHead tr,
                                                        float* __restrict__ rot_out, float* __restrict__ tr_out) {
  __shared__ float x[512];
  __shared__ float a1[2][128];
  __shared__ float a2[2][64];
  const int b = blockIdx.x, t = threadIdx.x;
  x[t] = h2[(long)b * 512 + t];
  x[t + 256] = h2[(long)b * 512 + 256 + t];
  __syncthreads();
  const int hd = t >> 7, u = t & 127;
  const MlpHead& H = hd ? tr : rot;
  {
    const float* w = H.w0 + (long)u * 512;
    float acc = 0.f;
    for (int k = 0; k < 512; ++k) acc += w[k] * x[k];
    a1[hd][u] = mishf_(acc + H.b0[u]);
  }
  __syncthreads();
  if (u < 64) {
    const float* w = H.w1 + (long)u * 128;
    float acc = 0.f;
    for (int k = 0; k < 128; ++k) acc += w[k] * a1[hd][k];
    a2[hd][u] = mishf_(acc + H.b1[u]);
  }
  __syncthreads();
  if (u < 3) {
    const float* w = H.w2 + (long)u * 64;
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc += w[k] * a2[hd][k];
    (hd ? tr_out : rot_out)[(long)b * 3 + u] = acc;
  }
}
void launch_mlp_heads(const float* h2, int B, MlpHead rot, MlpHead tr, float* rot_out, float* tr_out, hipStream_t st) {
  hipLaunchKernelGGL(mlp_heads_kernel, dim3(B), dim3(256), 0, st, h2, rot, tr, rot_out, tr_out);
  ATDN_HIP(hipGetLastError());
}

const float* zero_line() {
  static float* z = nullptr;
  if (!z) {
    ATDN_HIP(hipMalloc(&z, 256));
    ATDN_HIP(hipMemset(z, 0, 256));
    ATDN_HIP(hipDeviceSynchronize());
  }
  return z;
}

__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill(float* p, long n, float v, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<long>(cdivl(n, 256), 4096)), dim3(256), 0, st, p, n, v);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn

// ================================================================== split-f16 ("sf") variants
#include "sf.h"
#include <mutex>
#include <vector>
namespace atdn {

// ---- saturation counter of the sf format (sf.h)
namespace {
std::vector<void (*)(unsigned int*)>& sf_counter_setters() { static std::vector<void (*)(unsigned int*)> v; return v; }
std::mutex g_sf_counter_mutex;
struct SfCounterDev { int dev; unsigned int* ptr; };
std::vector<SfCounterDev> g_sf_counters;
unsigned int* sf_counter_for_current_device(bool create) {
  int dev = 0;
  ATDN_HIP(hipGetDevice(&dev));
  for (auto& c : g_sf_counters) if (c.dev == dev) return c.ptr;
  if (!create) return nullptr;
  unsigned int* p = nullptr;
  // p[0] = the counter, p[1] = the result slot of read-and-reset (allocated once: a hipMalloc / hipFree pair per read would
  // synchronise the whole device, every lane stream and the ingest stream, at each saturation check — ADVICE r4)
  ATDN_HIP(hipMalloc(&p, 2 * sizeof(unsigned int)));
  ATDN_HIP(hipMemset(p, 0, 2 * sizeof(unsigned int)));
  for (auto set : sf_counter_setters()) {   // one per translation unit that includes sf.h
    set(p);
    ATDN_HIP(hipGetLastError());
  }
  ATDN_HIP(hipDeviceSynchronize());
  g_sf_counters.push_back({dev, p});
  return p;
}
}  // namespace
void sf_counter_register(void (*setter)(unsigned int*)) { sf_counter_setters().push_back(setter); }
void sf_counter_attach() {
  std::lock_guard<std::mutex> lock(g_sf_counter_mutex);
  (void)sf_counter_for_current_device(true);
}
// read-and-reset is ONE atomic exchange executed on the caller's stream (behind everything the caller has launched there):
// a clamp that another stream's kernel records at any moment lands either in this read or in the next one, never between a
// read and a separate zeroing (the two-step form could lose it)
__global__ void sf_counter_exchange_kernel(unsigned int* counter, unsigned int* out) { *out = atomicExch(counter, 0u); }
unsigned int sf_counter_read_reset(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_sf_counter_mutex);
  unsigned int* p = sf_counter_for_current_device(true);
  unsigned int* slot = p + 1;   // (the mutex is held until the copy has landed: one reader at a time owns the slot)
  unsigned int v = 0;
  hipLaunchKernelGGL(sf_counter_exchange_kernel, dim3(1), dim3(1), 0, st, p, slot);
  ATDN_HIP(hipGetLastError());
  ATDN_HIP(hipMemcpyAsync(&v, slot, sizeof(v), hipMemcpyDeviceToHost, st));
  ATDN_HIP(hipStreamSynchronize(st));
  return v;
}

// one thread = 4 consecutive channels of one pixel: 16-B fp32 read, two 8-B half stores (hi plane, lo plane)
__device__ __forceinline__ void sf_store4(float* dst, long i4, float4 v) {
  // i4 = index of the float4 inside the fp32 tensor; group = i4 / 8, slot = i4 % 8 (4 channels each)
  const long grp = i4 >> 3;
  const int sl = (int)(i4 & 7);
  _Float16* g = reinterpret_cast<_Float16*>(dst + grp * 32) + sl * 4;
  const SfPair a = sf_split(v.x), b = sf_split(v.y), c = sf_split(v.z), d = sf_split(v.w);
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 hi = {a.hi, b.hi, c.hi, d.hi};
  h4 lo = {a.lo, b.lo, c.lo, d.lo};
  *reinterpret_cast<h4*>(g) = hi;
  *reinterpret_cast<h4*>(g + 32) = lo;
}
__device__ __forceinline__ float4 sf_load4(const float* src, long i4) {
  const long grp = i4 >> 3;
  const int sl = (int)(i4 & 7);
  const _Float16* g = reinterpret_cast<const _Float16*>(src + grp * 32) + sl * 4;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 hi = *reinterpret_cast<const h4*>(g);
  const h4 lo = *reinterpret_cast<const h4*>(g + 32);
  return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2],
                     (float)hi[3] + (float)lo[3]);
}

__global__ void to_sf_kernel(const float4* __restrict__ src, float* __restrict__ dst, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x)
    sf_store4(dst, i, src[i]);
}
void launch_to_sf(const float* src, float* dst, long rows, int C, hipStream_t st) {
  ATDN_CHECK(C % 32 == 0, "sf tensors need C % 32 == 0");
  const long total4 = rows * C / 4;
  hipLaunchKernelGGL(to_sf_kernel, dim3((unsigned)std::min<long>(cdivl(total4, 256), 4096)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(src), dst, total4);
  ATDN_HIP(hipGetLastError());
}
__global__ void from_sf_kernel(const float* __restrict__ src, float4* __restrict__ dst, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x)
    dst[i] = sf_load4(src, i);
}
void launch_from_sf(const float* src, float* dst, long rows, int C, hipStream_t st) {
  ATDN_CHECK(C % 32 == 0, "sf tensors need C % 32 == 0");
  const long total4 = rows * C / 4;
  hipLaunchKernelGGL(from_sf_kernel, dim3((unsigned)std::min<long>(cdivl(total4, 256), 4096)), dim3(256), 0, st, src,
                     reinterpret_cast<float4*>(dst), total4);
  ATDN_HIP(hipGetLastError());
}

__global__ void in_apply_sf_kernel(const float4* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ res,
                                   const float4* __restrict__ res_raw, const float* __restrict__ rmean,
                                   const float* __restrict__ rrstd, int res_relu, long per_img4, int C, long total4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long img = i / per_img4;
    const int c = (int)((i * 4) % C);
    const float* mu = mean + img * C + c;
    const float* rs = rstd + img * C + c;
    float4 v = x[i];
    v.x = fmaxf((v.x - mu[0]) * rs[0], 0.f);
    v.y = fmaxf((v.y - mu[1]) * rs[1], 0.f);
    v.z = fmaxf((v.z - mu[2]) * rs[2], 0.f);
    v.w = fmaxf((v.w - mu[3]) * rs[3], 0.f);
    if (res || res_raw) {
      float4 r;
      if (res) {
        r = sf_load4(res, i);
      } else {
        r = res_raw[i];
        const float* m2 = rmean + img * C + c;
        const float* r2 = rrstd + img * C + c;
        r.x = (r.x - m2[0]) * r2[0]; r.y = (r.y - m2[1]) * r2[1];
        r.z = (r.z - m2[2]) * r2[2]; r.w = (r.w - m2[3]) * r2[3];
        if (res_relu) {   // the shortcut is relu(IN(raw stem output)) (extractor.py:161-163), never materialised
          r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
        }
      }
      v.x = fmaxf(r.x + v.x, 0.f); v.y = fmaxf(r.y + v.y, 0.f);
      v.z = fmaxf(r.z + v.z, 0.f); v.w = fmaxf(r.w + v.w, 0.f);
    }
    sf_store4(y, i, v);
  }
}
void launch_in_apply_sf(const float* x, float* y, const float* mean, const float* rstd, const float* res,
                        const float* res_raw, const float* rmean, const float* rrstd, int nimg, long HW, int C,
                        hipStream_t st, bool res_relu) {
  ATDN_CHECK(C % 32 == 0, "sf tensors need C % 32 == 0");
  const long per_img4 = HW * C / 4, total4 = per_img4 * nimg;
  const int grid = (int)std::min<long>(cdivl(total4, 256), 256 * 16);
  hipLaunchKernelGGL(in_apply_sf_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(x), y, mean,
                     rstd, res, reinterpret_cast<const float4*>(res_raw), rmean, rrstd, res_relu ? 1 : 0, per_img4, C, total4);
  ATDN_HIP(hipGetLastError());
}

// 2x2 average of an sf feature map [img][H*W][C] -> [img][(H/2)*(W/2)][C] (floor sizes, as avg_pool2d(2, stride 2)).
// Correlation is linear in the target features, so <f1[p], mean of four f2> is the 2x2-pooled correlation of
// corr.py:28-30: level 1 of the pyramid then comes from a quarter-size GEMM instead of a pass over the 210 MB level 0.
__global__ void pool_features_sf_kernel(const float* __restrict__ src, int H, int W, int C, long sb, float* __restrict__ dst,
                                        long db, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Ho = H / 2, Wo = W / 2, c4 = C / 4;
  const int cq = (int)(i % c4);
  const long pq = i / c4;
  const int po = (int)(pq % ((long)Ho * Wo));
  const long img = pq / ((long)Ho * Wo);
  const int y = po / Wo, x = po - y * Wo;
  const float* s = src + img * sb;
  const long p00 = ((long)(2 * y) * W + 2 * x) * C;
  const float4 a = sf_load4(s, p00, 4 * cq), b = sf_load4(s, p00 + C, 4 * cq);
  const float4 c = sf_load4(s, p00 + (long)W * C, 4 * cq), d = sf_load4(s, p00 + (long)W * C + C, 4 * cq);
  const float4 o = make_float4((((a.x + b.x) + c.x) + d.x) * 0.25f, (((a.y + b.y) + c.y) + d.y) * 0.25f,
                               (((a.z + b.z) + c.z) + d.z) * 0.25f, (((a.w + b.w) + c.w) + d.w) * 0.25f);
  sf_store4(dst + img * db, (long)po * C, 4 * cq, o);
}
void launch_pool_features_sf(const float* src, int nimg, int H, int W, int C, long sb, float* dst, long db, hipStream_t st) {
  ATDN_CHECK(C % 32 == 0, "sf tensors need C % 32 == 0");
  const long total = (long)nimg * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(pool_features_sf_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, src, H, W, C, sb, dst, db, total);
  ATDN_HIP(hipGetLastError());
}

__global__ void init_coords_sf_kernel(const float* __restrict__ flow_init, int B, int H8, int W8,
                                      float* __restrict__ coords1, float* __restrict__ flow4, float* __restrict__ x,
                                      int ldx, int cflow) {
  const long N = (long)H8 * W8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const long img = i / N;
  const int m = (int)(i - img * N);
  const float x0 = (float)(m % W8), y0 = (float)(m / W8);
  float fx = 0.f, fy = 0.f;
  if (flow_init) { fx = flow_init[(img * 2 + 0) * N + m]; fy = flow_init[(img * 2 + 1) * N + m]; }
  const float cx = x0 + fx, cy = y0 + fy;
  coords1[i * 2 + 0] = cx;
  coords1[i * 2 + 1] = cy;
  const float flx = cx - x0, fly = cy - y0;
  reinterpret_cast<float4*>(flow4)[i] = make_float4(flx, fly, 0.f, 0.f);
  sf_store(x, i * ldx, cflow, flx);
  sf_store(x, i * ldx, cflow + 1, fly);
}
void launch_init_coords_sf(const float* flow_init, int B, int H8, int W8, float* coords1, float* flow4, float* x,
                           int ldx, int cflow, hipStream_t st) {
  const long n = (long)B * H8 * W8;
  hipLaunchKernelGGL(init_coords_sf_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, flow_init, B, H8, W8,
                     coords1, flow4, x, ldx, cflow);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
