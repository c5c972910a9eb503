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
__device__ __forceinline__ void sf_store4_flag(float* base, long off, int c, float4 v, bool& clamped) {
  // exact: the largest magnitude of the four against the limit (v_max3 + v_max with |.| source modifiers + one compare), and
  // two unordered compares for NaNs, which v_max drops. (Round 4 tested the SUM of the magnitudes here: four in-range values
  // of ~17000 each raised the alarm although nothing was clamped)
  const float lim = 65504.f;
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  clamped |= !(m <= lim) | __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);
  v.x = fminf(fmaxf(v.x, -lim), lim); v.y = fminf(fmaxf(v.y, -lim), lim);
  v.z = fminf(fmaxf(v.z, -lim), lim); v.w = fminf(fmaxf(v.w, -lim), lim);
  const SfPair a = sf_split_nocheck_(v.x), b = sf_split_nocheck_(v.y), d = sf_split_nocheck_(v.z), e = sf_split_nocheck_(v.w);
  _Float16* q = sf_ptr(base, off, c);
  *reinterpret_cast<f16x4*>(q) = f16x4{a.hi, b.hi, d.hi, e.hi};
  *reinterpret_cast<f16x4*>(q + 32) = f16x4{a.lo, b.lo, d.lo, e.lo};
}
__device__ __forceinline__ float4 sf_load4(const float* base, long off, int c) {
  const _Float16* q = sf_ptr(base, off, c);
  const f16x4 hi = *reinterpret_cast<const f16x4*>(q), lo = *reinterpret_cast<const f16x4*>(q + 32);
  return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2],
                     (float)hi[3] + (float)lo[3]);
}
__device__ __forceinline__ float sf_load(const float* base, long off, int c) {
  const _Float16* q = sf_ptr(base, off, c);
  return (float)q[0] + (float)q[32];
}

}  // namespace atdn
