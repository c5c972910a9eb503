// "sf" — split-f16 storage of fp32 activations/weights for the 3xf16 MFMA engine.
//
// x = hi + lo with hi = f16(x), lo = f16(x - hi): 22 significant bits, the same 4 bytes per element as fp32.
// Channels are stored in groups of 32: one group = 128 bytes = [32 x hi | 32 x lo], so a K-chunk of the
// implicit GEMM is still one contiguous 128-byte run per pixel and `ld`/channel offsets keep their fp32
// meaning (units of 4 bytes, channel offsets multiples of 32).
// Range: |x| <= 65504 (clamped). Below |x| ~ 0.06 the residual is an f16 subnormal and the absolute
// representation error floors at 3e-8 — fp32-epsilon of O(1) data, which is what flows through this network.
#pragma once
#include "common.h"

namespace atdn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct SfPair { _Float16 hi, lo; };

// Saturation counter. The format clamps at +-65504 (max finite f16): a value beyond that no longer round-trips, and a
// network whose activations get there (a hot BatchNorm-folded channel of a real checkpoint, say) must be noticed, not
// silently wrong. Every sf_split that clamps (or sees a NaN) bumps a device counter; atdn_gma_debug_read("sf_clamped")
// returns and resets it. The library is built without relocatable device code, so each translation unit carries its own
// pointer to the one counter; sf_counter_attach() (kernels.hip) points them all at it before the first launch.
namespace {
__device__ unsigned int* sf_clamp_counter_tu_ = nullptr;
// (a device variable with internal linkage cannot be reached by hipMemcpyToSymbol — the runtime looks symbols up by
// name — so every translation unit carries a one-thread kernel that sets its own copy)
__global__ void sf_counter_set_kernel_(unsigned int* p) { sf_clamp_counter_tu_ = p; }
void sf_counter_set_tu_(unsigned int* p) {
  hipLaunchKernelGGL(sf_counter_set_kernel_, dim3(1), dim3(1), 0, nullptr, p);
}
}  // namespace
void sf_counter_register(void (*setter)(unsigned int*));
namespace {
struct SfCounterRegistration { SfCounterRegistration() { sf_counter_register(&sf_counter_set_tu_); } };
static SfCounterRegistration sf_counter_registration_;
}
// points every translation unit's pointer at one device counter (idempotent per device); read_reset returns the count
void sf_counter_attach();
unsigned int sf_counter_read_reset(hipStream_t st);

// For loops that keep loads in flight: the clamp is remembered in a register and reported ONCE, after the loop
// (sf_report). A counted sf_split inside such a loop puts an atomic on a cold path of the loop, and the wait-count
// pass then drains every outstanding load at the loop header (it cannot know whether the cold path ran).
__device__ __forceinline__ SfPair sf_split_flag(float v, bool& clamped) {
  clamped |= !(fabsf(v) <= 65504.f);
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  SfPair p;
  p.hi = (_Float16)v;
  p.lo = (_Float16)(v - (float)p.hi);
  return p;
}
__device__ __forceinline__ void sf_report(bool clamped) {
  if (__builtin_expect(clamped, 0)) {
    unsigned int* c = *const_cast<unsigned int* volatile*>(&sf_clamp_counter_tu_);
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    if (c && (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u)) == 0u)
      atomicAdd(c, (unsigned)__builtin_popcountll(act));
  }
}

__device__ __forceinline__ SfPair sf_split(float v) {
  if (__builtin_expect(!(fabsf(v) <= 65504.f), 0)) {   // clamped value or NaN: rare, counted
    // one atomic per wave (the lanes that got here, counted by the first of them): a tensor that saturates everywhere
    // would otherwise serialise millions of atomics on one address
    // (volatile: the pointer is read HERE, on the cold path. Left to itself the compiler hoists the load above the
    // branch, and waiting for it — vmcnt retires in order — drains every prefetch a kernel has in flight, per call)
    unsigned int* c = *const_cast<unsigned int* volatile*>(&sf_clamp_counter_tu_);
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    if (c && (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u)) == 0u)
      atomicAdd(c, (unsigned)__builtin_popcountll(act));
  }
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  SfPair p;
  p.hi = (_Float16)v;
  p.lo = (_Float16)(v - (float)p.hi);
  return p;
}

// element (pixel offset `off` in 4-byte units, channel c) of an sf tensor
__device__ __forceinline__ _Float16* sf_ptr(float* base, long off, int c) {
  return reinterpret_cast<_Float16*>(base + off + (c & ~31)) + (c & 31);
}
__device__ __forceinline__ const _Float16* sf_ptr(const float* base, long off, int c) {
  return reinterpret_cast<const _Float16*>(base + off + (c & ~31)) + (c & 31);
}
__device__ __forceinline__ void sf_store(float* base, long off, int c, float v) {
  const SfPair p = sf_split(v);
  _Float16* q = sf_ptr(base, off, c);
  q[0] = p.hi;
  q[32] = p.lo;
}
// four consecutive channels c..c+3 (c % 4 == 0): one 8-byte access for the hi parts, one for the lo parts
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// (one saturation test for the four values — on the SUM of their magnitudes: it bounds every one of them, a NaN makes it a
// NaN and fails the `<=`, and a false alarm only takes the slower per-value path, which counts exactly — instead of four
// tests with a branch each; in range, the split is three conversions and a subtraction per value, no clamp)
__device__ __forceinline__ SfPair sf_split_nocheck_(float v) {
  SfPair p;
  p.hi = (_Float16)v;
  p.lo = (_Float16)(v - (float)p.hi);
  return p;
}
__device__ __forceinline__ void sf_store4(float* base, long off, int c, float4 v) {
  const float m = (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
  if (__builtin_expect(!(m <= 65504.f), 0)) {   // rare, counted: per value, as sf_split does
    const SfPair a = sf_split(v.x), b = sf_split(v.y), d = sf_split(v.z), e = sf_split(v.w);
    _Float16* q = sf_ptr(base, off, c);
    *reinterpret_cast<f16x4*>(q) = f16x4{a.hi, b.hi, d.hi, e.hi};
    *reinterpret_cast<f16x4*>(q + 32) = f16x4{a.lo, b.lo, d.lo, e.lo};
    return;
  }
  const SfPair a = sf_split_nocheck_(v.x), b = sf_split_nocheck_(v.y), d = sf_split_nocheck_(v.z), e = sf_split_nocheck_(v.w);
  _Float16* q = sf_ptr(base, off, c);
  *reinterpret_cast<f16x4*>(q) = f16x4{a.hi, b.hi, d.hi, e.hi};
  *reinterpret_cast<f16x4*>(q + 32) = f16x4{a.lo, b.lo, d.lo, e.lo};
}
// The same store for loops that keep memory operations in flight (convolution epilogues, attention x V, lookup): branch-free,
// the clamp goes into a register flag that the kernel reports once per wave after the loop (sf_report). sf_store4's counted
// cold path — a pointer load and an atomic behind a branch — makes the wait-count pass drain every outstanding store and
// prefetched operand at the join: the ConvGRU epilogues carried an s_waitcnt vmcnt(0) per pixel group (round 4).
// Round 5: the split and its inverse in the instructions gfx950 has for them. Two conversions round a PAIR to f16
// (v_cvt_pk_f16_f32), four v_fma_mix form the residuals lo = f16(v - hi) straight into the halves of two registers (fp32
// arithmetic on an f16 operand, one rounding: the same bits as convert-back, subtract, convert — tools/diag/sf_mix_check.hip):
// 10 vector instructions to store four values instead of 28 (conversions, subtractions, packing). The stamps of the ConvGRU
// kernels (tools/diag/conv_stamps.py) put a block's epilogue at 13-21 % of its life, bound by vector-instruction issue beside
// the partner wave's MFMAs.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sf_cvt_pk_(float a, float b) {
  const f32x2 x = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));   // v_cvt_pk_f16_f32, round to nearest even
}
__device__ __forceinline__ unsigned sf_residual_pk_(unsigned hp, float a, float b) {
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hp), "v"(b));
  return d;
}
__device__ __forceinline__ void sf_store4_flag(float* base, long off, int c, float4 v, bool& clamped) {
  // exact: the largest magnitude of the four against the limit (v_max3 + v_max with |.| source modifiers + one compare), and
  // two unordered compares for NaNs, which v_max drops. (Round 4 tested the SUM of the magnitudes here: four in-range values
  // of ~17000 each raised the alarm although nothing was clamped)
  const float lim = 65504.f;
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  clamped |= !(m <= lim) | __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);
  v.x = __builtin_amdgcn_fmed3f(v.x, -lim, lim); v.y = __builtin_amdgcn_fmed3f(v.y, -lim, lim);
  v.z = __builtin_amdgcn_fmed3f(v.z, -lim, lim); v.w = __builtin_amdgcn_fmed3f(v.w, -lim, lim);
  const unsigned h0 = sf_cvt_pk_(v.x, v.y), h1 = sf_cvt_pk_(v.z, v.w);
  const unsigned l0 = sf_residual_pk_(h0, v.x, v.y), l1 = sf_residual_pk_(h1, v.z, v.w);
  _Float16* q = sf_ptr(base, off, c);
  *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(q + 32) = make_uint2(l0, l1);
}
// hi + lo: convert, convert, add. (ONE v_fma_mix_f32 per value — hi * 1 + lo in fp32 — gives the same bits,
// tools/diag/sf_mix_check.hip, and was measured SLOWER in every kernel whose epilogue decodes an operand: cnet +0.24 ms, the z|r
// gates +0.06 ms per pass and forward, profiles/r05_ab_sf_ops.txt. The residual direction, v_fma_mixlo / mixhi, is neutral to
// slightly faster and stays.)
__device__ __forceinline__ float sf_join_lo_(unsigned hi, unsigned lo) {
  return (float)__builtin_bit_cast(f16x2, hi)[0] + (float)__builtin_bit_cast(f16x2, lo)[0];
}
__device__ __forceinline__ float sf_join_hi_(unsigned hi, unsigned lo) {
  return (float)__builtin_bit_cast(f16x2, hi)[1] + (float)__builtin_bit_cast(f16x2, lo)[1];
}
__device__ __forceinline__ float4 sf_load4(const float* base, long off, int c) {
  const _Float16* q = sf_ptr(base, off, c);
  const uint2 hi = *reinterpret_cast<const uint2*>(q), lo = *reinterpret_cast<const uint2*>(q + 32);
  return make_float4(sf_join_lo_(hi.x, lo.x), sf_join_hi_(hi.x, lo.x), sf_join_lo_(hi.y, lo.y), sf_join_hi_(hi.y, lo.y));
}
// The same two with the address as a UNIFORM base (a scalar register pair: the image's slice of the tensor) plus an unsigned
// 32-bit element offset: one vector add per access (global_load / global_store with a scalar base) instead of the 64-bit
// vector arithmetic a `long` offset costs per access (2.4 vector instructions per stored value in the q gate's epilogue).
// Every tensor of this library is < 4 GB per image slice.
__device__ __forceinline__ unsigned sf_byte_offset_(unsigned off, int c) {
  return 4u * (off + (unsigned)(c & ~31)) + 2u * (unsigned)(c & 31);
}
__device__ __forceinline__ void sf_store4_flag_u(float* base, unsigned off, int c, float4 v, bool& clamped) {
  const float lim = 65504.f;
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  clamped |= !(m <= lim) | __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);
  v.x = __builtin_amdgcn_fmed3f(v.x, -lim, lim); v.y = __builtin_amdgcn_fmed3f(v.y, -lim, lim);
  v.z = __builtin_amdgcn_fmed3f(v.z, -lim, lim); v.w = __builtin_amdgcn_fmed3f(v.w, -lim, lim);
  const unsigned h0 = sf_cvt_pk_(v.x, v.y), h1 = sf_cvt_pk_(v.z, v.w);
  const unsigned l0 = sf_residual_pk_(h0, v.x, v.y), l1 = sf_residual_pk_(h1, v.z, v.w);
  char* q = reinterpret_cast<char*>(base) + sf_byte_offset_(off, c);
  *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(q + 64) = make_uint2(l0, l1);
}
__device__ __forceinline__ float4 sf_load4u(const float* base, unsigned off, int c) {
  const char* q = reinterpret_cast<const char*>(base) + sf_byte_offset_(off, c);
  const uint2 hi = *reinterpret_cast<const uint2*>(q), lo = *reinterpret_cast<const uint2*>(q + 64);
  return make_float4(sf_join_lo_(hi.x, lo.x), sf_join_hi_(hi.x, lo.x), sf_join_lo_(hi.y, lo.y), sf_join_hi_(hi.y, lo.y));
}
__device__ __forceinline__ float sf_load(const float* base, long off, int c) {
  const _Float16* q = sf_ptr(base, off, c);
  return (float)q[0] + (float)q[32];
}

}  // namespace atdn
