// Shared host/device helpers for libatdn_hip (gfx950 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace atdn {

void set_last_error(const std::string& msg);

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define ATDN_HIP(expr)                                                                      \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      char _b[512];                                                                         \
      snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw ::atdn::Error(_b);                                                              \
    }                                                                                       \
  } while (0)

#define ATDN_CHECK(cond, msg)                                                               \
  do {                                                                                      \
    if (!(cond)) {                                                                          \
      char _b[512];                                                                         \
      snprintf(_b, sizeof _b, "%s [%s] (%s:%d)", msg, #cond, __FILE__, __LINE__);           \
      throw ::atdn::Error(_b);                                                              \
    }                                                                                       \
  } while (0)

// makes `dev` current for a scope (destructors of handles that own memory on ONE device: the caller's current device
// may be another one — multi-GPU processes)
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline long cdivl(long a, long b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// epilogues that take their per-column constants as an argument (struct Col, col(n), store_c)
template <class E, class = void> struct epi_bias_arg : std::false_type {};
template <class E> struct epi_bias_arg<E, std::void_t<typename E::Col>> : std::true_type {};
// its Col type (an empty struct for epilogues without one)
template <class E, class = void> struct EpiCol { struct type {}; };
template <class E> struct EpiCol<E, std::void_t<typename E::Col>> { using type = typename E::Col; };

// Bijective XCD-aware remap of a 1-D block id: blocks that land on the same XCD
// (observed round-robin: id % 8) receive a contiguous range of logical ids, so
// neighbouring tiles share that XCD's L2. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// value select on a float4 (a ternary on the struct type becomes a pointer select + memcpy and drags every
// prefetch array into scratch memory)
__device__ __forceinline__ float4 keep_if(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// Gate functions of the split-f16 ConvGRU epilogues: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the library expf / tanhf
// and an IEEE division (~20 and ~35 vector instructions per value: 768 values per pixel and iteration, in an epilogue that no
// MFMA overlaps). sigmoid: relative error <= ~3e-7. tanh = sign(x) (1 - 2 / (1 + e^(2|x|))): ABSOLUTE error <= ~1.5e-7 (near
// zero the subtraction cancels; the value enters h' = (1 - z) h + z q, where only the absolute error counts).
__device__ __forceinline__ float sigmoid_fast_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float tanh_fast_(float x) {
  const float t = __builtin_amdgcn_exp2f(fabsf(x) * 2.8853900817779268f);
  return copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + t), x);
}
__device__ __forceinline__ float mishf_(float x) {
  const float sp = (x > 20.0f) ? x : log1pf(expf(x));
  return x * tanhf(sp);
}

}  // namespace atdn
