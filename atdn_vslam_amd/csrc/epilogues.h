// Fused epilogues of the implicit-GEMM engine: called once per valid output
// element as ep(img, pix, n, acc). All tensors NHWC; `ob` = per-image stride.
#pragma once
#include <type_traits>
#include "common.h"

namespace atdn {

enum { ACT_NONE = 0, ACT_RELU = 1 };

// y = act(acc + bias)            (cnet convs with BatchNorm folded into w/b; motion encoder; flow/mask heads)
template <int ACT>
struct EpiBias {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  const float* bias;  // may be null
  float* dst; long ob; int ld;
  float scale;        // applied after bias (mask head: 0.25), 1 otherwise
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const { store_c(img, m, n, a, col(n)); }
  // element form with the per-column constants passed in (fetched once per output column by the kernel, see SfBias)
  struct Col { float b; };
  __device__ __forceinline__ Col col(int n) const { return {bias ? bias[n] : 0.f}; }
  __device__ __forceinline__ void store_c(int img, int m, int n, float a, Col c) const {
    float v = a + c.b;
    if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
    dst[(long)img * ob + (long)m * ld + n] = v * scale;
  }
  // channel-vector form (conv_sf6.h): channels n..n+3 of pixel m; 3x3 and 1x5 / 5x1 halo kernels
  static constexpr bool kVec4 = true;
  static constexpr int kGen6 = 3;
  // bias of channels n..n+3, fetched once per channel run by the kernel (see SfBias)
  __device__ __forceinline__ float4 bias4(int n) const {
    return bias ? make_float4(bias[n], bias[n + 1], bias[n + 2], bias[n + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void store4(int img, int m, int n, float4 a, float4 b, bool& /*clamped: sf epilogues only*/) const {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    if (ACT == ACT_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    float* d = dst + (long)img * ob + (long)m * ld + n;
    if ((ld & 3) == 0 && (ob & 3) == 0) {
      *reinterpret_cast<float4*>(d) = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
    } else {
      d[0] = a.x * scale; d[1] = a.y * scale; d[2] = a.z * scale; d[3] = a.w * scale;
    }
  }
};

// raw = acc + bias, plus per-(32-row group, channel) partial statistics for InstanceNorm (fnet)
struct EpiBiasStats {
  static constexpr bool kStats = true;
  static constexpr bool kPrefetch = false;
  const float* bias;
  float* dst; long ob; int ld;
  float* part_sum; float* part_m2; int groups_per_img;
  float* part_cnt = nullptr;  // valid rows per group (only the 2-D tiled kernel writes it)
  static constexpr int kGen6 = 1;  // conv_sf6.h, classic (pixel-major) orientation
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const { store_c(img, m, n, a, col(n)); }
  // per-column constant passed in (see SfBias): one bias load per output column, not one between every two stores
  struct Col { float b; };
  __device__ __forceinline__ Col col(int n) const { return {bias[n]}; }
  __device__ __forceinline__ void store_c(int img, int m, int n, float a, Col c) const {
    dst[(long)img * ob + (long)m * ld + n] = a + c.b;
  }
};

// out = relu(res + relu(acc + bias))      (cnet residual block tail, extractor.py:47-55 with folded BN)
struct EpiBiasReluAddRelu {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;  // load(): issued for all 16 elements of a tile before any apply()
  const float* bias;
  const float* res; long rb; int ldr;
  float* dst; long ob; int ld;
  struct Aux { float r; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {res[(long)img * rb + (long)m * ldr + n]}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float y = fmaxf(a + bias[n], 0.f);
    dst[(long)img * ob + (long)m * ld + n] = fmaxf(x.r + y, 0.f);
  }
};

// cnet head: channels [0,128) -> tanh -> hidden state; [128,256) -> relu -> context slice of x  (network.py:94-97)
struct EpiContextSplit {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  const float* bias;
  float* net; long nb;            // [img][pix][128]
  float* inp; long ib; int ldi;   // x buffer, channels [0,128)
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const float v = a + bias[n];
    if (n < 128) net[(long)img * nb + (long)m * 128 + n] = tanhf(v);
    else inp[(long)img * ib + (long)m * ldi + (n - 128)] = fmaxf(v, 0.f);
  }
};

// y = acc * scale     (all-pairs correlation /sqrt(C), corr.py:63; attention logits)
struct EpiScale {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  float scale;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    dst[(long)img * ob + (long)m * ld + n] = a * scale;
  }
};

// to_qk (gma.py:57-60): q columns [0,nq) are scaled before the dot product
struct EpiQK {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  float scale; int nq;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    dst[(long)img * ob + (long)m * ld + n] = (n < nq) ? a * scale : a;
  }
};

// transposed store: dst[img][n][m]   (to_v output becomes the K-contiguous "weight" of attention·V)
struct EpiStoreT {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    dst[(long)img * ob + (long)n * ld + m] = a;
  }
};

// out = mf + gamma * acc    (gma.py:113)
struct EpiAggregate {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* gamma;  // device scalar
  const float* mf; long mb; int ldm;
  float* dst; long ob; int ld;
  struct Aux { float x; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {mf[(long)img * mb + (long)m * ldm + n]}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    dst[(long)img * ob + (long)m * ld + n] = x.x + gamma[0] * a;
  }
};

// fused z‖r convolution of the separable ConvGRU (update.py:50-52,57-59):
//   n <  128: z = sigmoid(.) -> zbuf ;  n >= 128: r = sigmoid(.), store r*h
struct EpiGruZR {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;  // [256] = bz ‖ br
  const float* h;     // [img][pix][128]
  float* z; float* rh; long ob;
  struct Aux { float h; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const {
    return {h[(long)img * ob + (long)m * 128 + (n & 127)]};
  }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float v = sigmoidf_(a + bias[n]);
    const long o = (long)img * ob + (long)m * 128;
    if (n < 128) z[o + n] = v;
    else rh[o + n - 128] = v * x.h;
  }
};

// q convolution + state update (update.py:53-54,60-61): h' = (1-z) h + z tanh(.)
struct EpiGruQ {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;
  const float* h; const float* z;
  float* hout; long ob;
  struct Aux { float h, z; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const {
    const long o = (long)img * ob + (long)m * 128 + n;
    return {h[o], z[o]};
  }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float q = tanhf(a + bias[n]);
    hout[(long)img * ob + (long)m * 128 + n] = __builtin_fmaf(x.z, q, (1.f - x.z) * x.h);   // one explicit fma: see SfGruQ::blend
  }
};

// flow head tail (update.py:15, network.py:116,111): coords1 += delta; flow = coords1 - coords0
struct EpiFlowDelta {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;  // [2]
  float* coords1;     // [img][pix][2]
  float* flow4;       // [img][pix][4] (x, y, 0, 0): input of convf1
  float* xflow; int ldx; long xb;  // flow channels inside the GRU input x (motion_features[126:128])
  int W8; long npix;
  struct Aux { float c; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {coords1[((long)img * npix + m) * 2 + n]}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const long p = (long)img * npix + m;
    const float c1 = x.c + (a + bias[n]);
    coords1[p * 2 + n] = c1;
    const float c0 = (n == 0) ? (float)(m % W8) : (float)(m / W8);
    const float f = c1 - c0;
    flow4[p * 4 + n] = f;
    xflow[(long)img * xb + (long)m * ldx + n] = f;
  }
};

// CLVO `Conv` block tail (layers/conv.py:37): bn(mish(acc + bias)) with eval-BN as scale/shift
struct EpiMishBN {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  const float* bias; const float* sc; const float* sh;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const { store_c(img, m, n, a, col(n)); }
  struct Col { float b, sc, sh; };
  __device__ __forceinline__ Col col(int n) const { return {bias[n], sc[n], sh[n]}; }
  __device__ __forceinline__ void store_c(int img, int m, int n, float a, Col c) const {
    dst[(long)img * ob + (long)m * ld + n] = mishf_(a + c.b) * c.sc + c.sh;
  }
};

// CLVO `ResidualConv` tail (layers/conv.py:83-90): bn_out(mish(bn_b(mish(acc + bias)) + skip))
struct EpiMishBNSkipMishBN {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias; const float* sc1; const float* sh1;
  const float* skip; long sb; int lds;
  const float* sc2; const float* sh2;
  float* dst; long ob; int ld;
  struct Aux { float s; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {skip[(long)img * sb + (long)m * lds + n]}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float y = mishf_(a + bias[n]) * sc1[n] + sh1[n];
    dst[(long)img * ob + (long)m * ld + n] = mishf_(y + x.s) * sc2[n] + sh2[n];
  }
};

}  // namespace atdn
