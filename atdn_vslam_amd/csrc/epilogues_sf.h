// Epilogues of the split-f16 engine. Same contract as epilogues.h (ep(img, pix, n, acc), acc already scaled
// by the layer's weight scale); tensors that feed another MFMA kernel are written in sf format (sf.h),
// tensors read here that were produced in sf format are decoded on the fly.
#pragma once
#include "epilogues.h"
#include "sf.h"

namespace atdn {

// y = act(acc + bias) -> sf
template <int ACT>
struct SfBias {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  const float* bias;  // may be null
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    float v = a + (bias ? bias[n] : 0.f);
    if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
    sf_store(dst, (long)img * ob + (long)m * ld, n, v);
  }
};

// out = relu(res + relu(acc + bias)), res and out in sf
struct SfBiasReluAddRelu {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;
  const float* res; long rb; int ldr;
  float* dst; long ob; int ld;
  struct Aux { float r; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {sf_load(res, (long)img * rb + (long)m * ldr, n)}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float y = fmaxf(a + bias[n], 0.f);
    sf_store(dst, (long)img * ob + (long)m * ld, n, fmaxf(x.r + y, 0.f));
  }
};

struct SfContextSplit {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  const float* bias;
  float* net; long nb;            // sf [img][pix][128]
  float* inp; long ib; int ldi;   // sf x buffer, channels [0,128)
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const float v = a + bias[n];
    if (n < 128) sf_store(net, (long)img * nb + (long)m * 128, n, tanhf(v));
    else sf_store(inp, (long)img * ib + (long)m * ldi, n - 128, fmaxf(v, 0.f));
  }
};

struct SfQK {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  float scale; int nq;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    sf_store(dst, (long)img * ob + (long)m * ld, n, (n < nq) ? a * scale : a);
  }
};

// transposed sf store: dst[img][n][m]  (m is the K index of attention·V)
struct SfStoreT {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    sf_store(dst, (long)img * ob + (long)n * ld, m, a);
  }
};

struct SfAggregate {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* gamma;
  const float* mf; long mb; int ldm;   // sf
  float* dst; long ob; int ld;          // sf
  struct Aux { float x; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {sf_load(mf, (long)img * mb + (long)m * ldm, n)}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    sf_store(dst, (long)img * ob + (long)m * ld, n, x.x + gamma[0] * a);
  }
};

struct SfGruZR {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;
  const float* h;   // sf [img][pix][128]
  float* z;         // fp32 (only the q epilogue reads it)
  float* rh;        // sf
  long ob;
  const float* pre; long pb;  // fp32 [img][pix][256]: the iteration-invariant context-channel part of the conv
  struct Aux { float h, p; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const {
    return {sf_load(h, (long)img * ob + (long)m * 128, n & 127), pre[(long)img * pb + (long)m * 256 + n]};
  }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float v = sigmoidf_((a + x.p) + bias[n]);
    const long o = (long)img * ob + (long)m * 128;
    if (n < 128) z[o + n] = v;
    else sf_store(rh, o, n - 128, v * x.h);
  }
};

struct SfGruQ {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;
  const float* h;   // sf
  const float* z;   // fp32
  float* hout;      // sf
  long ob;
  const float* pre;  // fp32 [img][pix][128], same per-image stride as h
  struct Aux { float h, z, p; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const {
    const long o = (long)img * ob + (long)m * 128;
    return {sf_load(h, o, n), z[o + n], pre[o + n]};
  }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float q = tanhf((a + x.p) + bias[n]);
    sf_store(hout, (long)img * ob + (long)m * 128, n, (1.f - x.z) * x.h + x.z * q);
  }
};

struct SfFlowDelta {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = true;
  const float* bias;
  float* coords1;   // fp32 [img][pix][2]
  float* flow4;     // fp32 [img][pix][4]
  float* x; int ldx; long xb; int cflow;  // sf GRU input, flow channels cflow, cflow+1
  int W8; long npix;
  struct Aux { float c; };
  __device__ __forceinline__ Aux load(int img, int m, int n) const { return {coords1[((long)img * npix + m) * 2 + n]}; }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux xx) const {
    const long p = (long)img * npix + m;
    const float c1 = xx.c + (a + bias[n]);
    coords1[p * 2 + n] = c1;
    const float c0 = (n == 0) ? (float)(m % W8) : (float)(m / W8);
    const float f = c1 - c0;
    flow4[p * 4 + n] = f;
    sf_store(x, (long)img * xb + (long)m * ldx, cflow + n, f);
  }
};

}  // namespace atdn
