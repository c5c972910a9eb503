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
  // field-wise choice between two epilogues of a pair launch (conv_sf6_pair_kernel): uniform selects, everything stays scalar —
  // indexing an array of kernel arguments instead makes the compiler spill the array to LDS, 90 bytes per thread
  static __device__ __forceinline__ SfBias pick(const SfBias& a, const SfBias& b, bool second) {
    return SfBias{second ? b.bias : a.bias, second ? b.dst : a.dst, second ? b.ob : a.ob, second ? b.ld : a.ld};
  }
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const { store_c(img, m, n, a, col(n)); }
  // element form with the per-column constants passed in: kernels fetch col(n) once per output column instead of
  // once per element (a load between the stores of consecutive rows makes each row wait for the previous row's stores)
  struct Col { float b; };
  __device__ __forceinline__ Col col(int n) const { return {bias ? bias[n] : 0.f}; }
  __device__ __forceinline__ void store_c(int img, int m, int n, float a, Col c) const {
    float v = a + c.b;
    if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
    sf_store(dst, (long)img * ob + (long)m * ld, n, v);
  }
  // channel-vector form (conv_sf6.h): channels n..n+3 of pixel m
  static constexpr bool kVec4 = true;
  static constexpr int kGen6 = 1;  // conv_sf6.h kernel shapes: 1 = 3x3, 2 = 1x5 / 5x1
  // bias of channels n..n+3, fetched ONCE per channel run by the kernel (a load inside store4 sits between the
  // stores of consecutive pixels and its wait, vmcnt(0), also waits for those stores)
  __device__ __forceinline__ float4 bias4(int n) const {
    return bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // (`clamped`: the kernel's saturation flag, reported once after its loops — sf.h, sf_store4_flag)
  // (Round 5 measured this store with the RAW accumulator and the weight scale folded into the bias addition as an FMA, and
  // with a scalar base + 32-bit offset address, as the ConvGRU gates now have it (SfGruZR): two fewer vector instructions per
  // value, but the 128- and 64-wide 3x3 kernels came out at 276 / 188 registers instead of 244 / 168 — one wave per SIMD less —
  // and cnet / the motion encoder lost 0.1 / 0.3 ms per forward. profiles/r05_ab_sf_ops.txt)
  __device__ __forceinline__ void store4(int img, int m, int n, float4 a, float4 b, bool& clamped) const {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    if (ACT == ACT_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    sf_store4_flag(dst, (long)img * ob + (long)m * ld, n, a, clamped);
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
  static constexpr bool kVec4 = true;
  static constexpr int kGen6 = 1;  // conv_sf6.h kernel shapes: 1 = 3x3, 2 = 1x5 / 5x1
  struct Aux4 { float4 r; };
  __device__ __forceinline__ Aux4 load4(int img, int m, int n) const { return {sf_load4(res, (long)img * rb + (long)m * ldr, n)}; }
  __device__ __forceinline__ float4 bias4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
  __device__ __forceinline__ void apply4(int img, int m, int n, float4 a, Aux4 x, float4 b, bool& clamped) const {
    float4 o;
    o.x = fmaxf(x.r.x + fmaxf(a.x + b.x, 0.f), 0.f);
    o.y = fmaxf(x.r.y + fmaxf(a.y + b.y, 0.f), 0.f);
    o.z = fmaxf(x.r.z + fmaxf(a.z + b.z, 0.f), 0.f);
    o.w = fmaxf(x.r.w + fmaxf(a.w + b.w, 0.f), 0.f);
    sf_store4_flag(dst, (long)img * ob + (long)m * ld, n, o, clamped);
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

// V^T for attention x V (attention.hip): dst[img][channel m][key n] in sf, with the 32 keys of a chunk stored in the order
// the consumer's MFMA operand wants them — slot g (16 bytes) = keys 4 g + (i & 3) + 16 (i >> 2), i = 0..7 (attention.h) —
// so the consumer copies 16-byte slots verbatim into its LDS image (round 3: it used to move 8-byte pieces to permuted
// positions, ds_write_b64 with 2-way bank conflicts at the 160-byte row pitch).
struct SfVT {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  float* dst; long ob; int ld;
  __device__ __forceinline__ static int perm(int n) {   // key n -> its position inside the chunk
    const int k = n & 31;
    return (n & ~31) | (((k >> 2) & 3) << 3) | ((k >> 4) << 2) | (k & 3);
  }
  __device__ __forceinline__ void operator()(int img, int m, int n, float a) const {
    sf_store(dst, (long)img * ob + (long)m * ld, perm(n), a);
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
    const float v = sigmoid_fast_((a + x.p) + bias[n]);
    const long o = (long)img * ob + (long)m * 128;
    if (n < 128) z[o + n] = v;
    else sf_store(rh, o, n - 128, v * x.h);
  }
  static constexpr bool kVec4 = true;
  static constexpr int kGen6 = 2;  // conv_sf6.h kernel shapes: 1 = 3x3, 2 = 1x5 / 5x1
  struct Aux4 { float4 h, p; };
  __device__ __forceinline__ Aux4 load4(int img, int m, int n) const {
    return {sf_load4u(h + (long)img * ob, (unsigned)m * 128u, n & 127),
            *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pre + (long)img * pb) + 4u * ((unsigned)m * 256u + (unsigned)n))};
  }
  __device__ __forceinline__ float4 bias4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
  // kRawAcc (round 5): `a` is the RAW accumulator and `ws` the layer's weight scale, a power of two (weights.h) — so a * ws is
  // exact and fma(a, ws, p) has the bits of (a * ws) + p: the scaling multiply per value is gone from the epilogue. Addresses:
  // the image's slice as a scalar base + an unsigned 32-bit offset (sf.h). Both gates together: -0.3 ms per forward
  // (profiles/r05_ab_sf_ops.txt); the plain-store epilogues lost registers to the same change and keep the old form (SfBias).
  static constexpr bool kRawAcc = true;
  static constexpr bool kRing3 = true;   // conv_sf6.h: three weight slots (K = 12 chunks x 5 taps)
  __device__ __forceinline__ void apply4(int img, int m, int n, float4 a, Aux4 x, float4 b, bool& clamped, float ws) const {
    float4 v;
    v.x = sigmoid_fast_(__builtin_fmaf(a.x, ws, x.p.x) + b.x);
    v.y = sigmoid_fast_(__builtin_fmaf(a.y, ws, x.p.y) + b.y);
    v.z = sigmoid_fast_(__builtin_fmaf(a.z, ws, x.p.z) + b.z);
    v.w = sigmoid_fast_(__builtin_fmaf(a.w, ws, x.p.w) + b.w);
    const unsigned o = (unsigned)m * 128u;
    if (n < 128) *reinterpret_cast<float4*>(reinterpret_cast<char*>(z + (long)img * ob) + 4u * (o + (unsigned)n)) = v;
    else sf_store4_flag_u(rh + (long)img * ob, o, n - 128, make_float4(v.x * x.h.x, v.y * x.h.y, v.z * x.h.z, v.w * x.h.w), clamped);
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
  // h' = (1 - z) h + z q, written as ONE explicit fma on a rounded product: "a*b + c*d" leaves the compiler the choice of which
  // product to fuse, and it chose differently in two instantiations of the same kernel (64- and 128-wide blocks) once the
  // surrounding code changed — a pair then came out differently alone and inside an 8-pair launch
  // (tests/test_gpu_parity.py: test_large_batch_tile_path_matches_single_pairs, round 4)
  __device__ __forceinline__ static float blend(float z, float h, float q) { return __builtin_fmaf(z, q, (1.f - z) * h); }
  __device__ __forceinline__ void apply(int img, int m, int n, float a, Aux x) const {
    const float q = tanh_fast_((a + x.p) + bias[n]);
    sf_store(hout, (long)img * ob + (long)m * 128, n, blend(x.z, x.h, q));
  }
  static constexpr bool kVec4 = true;
  static constexpr int kGen6 = 2;  // conv_sf6.h kernel shapes: 1 = 3x3, 2 = 1x5 / 5x1
  struct Aux4 { float4 h, z, p; };
  __device__ __forceinline__ Aux4 load4(int img, int m, int n) const {
    const long ib = (long)img * ob;
    const unsigned o = 4u * ((unsigned)m * 128u + (unsigned)n);
    return {sf_load4u(h + ib, (unsigned)m * 128u, n), *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(z + ib) + o),
            *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pre + ib) + o)};
  }
  __device__ __forceinline__ float4 bias4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
  static constexpr bool kRawAcc = true;   // see SfGruZR
  static constexpr bool kRing3 = true;
  __device__ __forceinline__ void apply4(int img, int m, int n, float4 a, Aux4 x, float4 b, bool& clamped, float ws) const {
    float4 o;
    o.x = blend(x.z.x, x.h.x, tanh_fast_(__builtin_fmaf(a.x, ws, x.p.x) + b.x));
    o.y = blend(x.z.y, x.h.y, tanh_fast_(__builtin_fmaf(a.y, ws, x.p.y) + b.y));
    o.z = blend(x.z.z, x.h.z, tanh_fast_(__builtin_fmaf(a.z, ws, x.p.z) + b.z));
    o.w = blend(x.z.w, x.h.w, tanh_fast_(__builtin_fmaf(a.w, ws, x.p.w) + b.w));
    sf_store4_flag_u(hout + (long)img * ob, (unsigned)m * 128u, n, o, clamped);
  }
};

struct SfFlowDelta {
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

// Flow head, both convolutions in one pass over the hidden state (update.py:8-16: conv2(relu(conv1(h)))). conv1's
// 256-channel output of a pixel never leaves the block: every wave multiplies relu(conv1 + bias) of its 32 channels with
// conv2's weights for the 9 taps x 2 outputs on the matrix engine (conv_sf6.h, the kFlowHead branch of the channel-vector
// epilogue: split-f16 products, conv2's weights pre-scaled by the power of two `w2mul`), the block adds its waves' 18 partial
// sums per pixel in fixed order and writes G[pixel][tap * 2 + output]; a gather over each pixel's 3 x 3 neighbourhood
// (small_convs.hip: flow_gather_kernel) finishes conv2 and applies the flow update. Saves the 58 MB round trip of conv1's
// output per iteration (8 pairs) and the separate conv2 kernel.
struct SfFlowHeadPartial {
  static constexpr bool kStats = false;
  static constexpr bool kPrefetch = false;
  static constexpr bool kVec4 = true;
  static constexpr bool kFlowHead = true;
  static constexpr int kGen6 = 1;
  const float* bias;   // conv1 bias [256]
  const float* w2;     // conv2 weights, fp32 [18][256]: row tap * 2 + output
  float* G; long npix; // [channel block of the launch][img][pix][18]: a block of BN < 256 channels writes its own copy, the gather adds them
  float w2mul, w2inv;  // power-of-two scale of conv2's weights for the split-f16 product (max |w| in [1, 2)) and its inverse
  long gstride = 0;    // floats between the copies of G (0: one 256-wide block per pixel tile)
  __device__ __forceinline__ float4 bias4(int n) const { return *reinterpret_cast<const float4*>(bias + n); }
  __device__ __forceinline__ void store4(int, int, int, float4, float4, bool&) const {}
  __device__ __forceinline__ void operator()(int, int, int, float) const {}
};

}  // namespace atdn
