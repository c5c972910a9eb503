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
  const float* bias;
  const float* res; long rb; int ldr;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const float y = fmaxf(a + bias[n], 0.f);
    const float r = sf_load(res, (long)img * rb + (long)m * ldr, n);
    sf_store(dst, (long)img * ob + (long)m * ld, n, fmaxf(r + y, 0.f));
  }
};

struct SfContextSplit {
  static constexpr bool kStats = false;
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
  float scale; int nq;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    sf_store(dst, (long)img * ob + (long)m * ld, n, (n < nq) ? a * scale : a);
  }
};

// transposed sf store: dst[img][n][m]  (m is the K index of attention·V)
struct SfStoreT {
  static constexpr bool kStats = false;
  float* dst; long ob; int ld;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    sf_store(dst, (long)img * ob + (long)n * ld, m, a);
  }
};

struct SfAggregate {
  static constexpr bool kStats = false;
  const float* gamma;
  const float* mf; long mb; int ldm;   // sf
  float* dst; long ob; int ld;          // sf
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const float x = sf_load(mf, (long)img * mb + (long)m * ldm, n);
    sf_store(dst, (long)img * ob + (long)m * ld, n, x + gamma[0] * a);
  }
};

struct SfGruZR {
  static constexpr bool kStats = false;
  const float* bias;
  const float* h;   // sf [img][pix][128]
  float* z;         // fp32 (only the q epilogue reads it)
  float* rh;        // sf
  long ob;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const float v = sigmoidf_(a + bias[n]);
    const long o = (long)img * ob + (long)m * 128;
    if (n < 128) z[o + n] = v;
    else sf_store(rh, o, n - 128, v * sf_load(h, o, n - 128));
  }
};

struct SfGruQ {
  static constexpr bool kStats = false;
  const float* bias;
  const float* h;   // sf
  const float* z;   // fp32
  float* hout;      // sf
  long ob;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const float q = tanhf(a + bias[n]);
    const long o = (long)img * ob + (long)m * 128;
    const float zz = z[o + n];
    sf_store(hout, o, n, (1.f - zz) * sf_load(h, o, n) + zz * q);
  }
};

struct SfFlowDelta {
  static constexpr bool kStats = false;
  const float* bias;
  float* coords1;   // fp32 [img][pix][2]
  float* flow4;     // fp32 [img][pix][4]
  float* x; int ldx; long xb; int cflow;  // sf GRU input, flow channels cflow, cflow+1
  int W8; long npix;
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    const long p = (long)img * npix + m;
    const float c1 = coords1[p * 2 + n] + (a + bias[n]);
    coords1[p * 2 + n] = c1;
    const float c0 = (n == 0) ? (float)(m % W8) : (float)(m / W8);
    const float f = c1 - c0;
    flow4[p * 4 + n] = f;
    sf_store(x, (long)img * xb + (long)m * ldx, cflow + n, f);
  }
};

}  // namespace atdn
