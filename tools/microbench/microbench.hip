// Diagnostic micro-benchmarks (ablation builds of the product kernels: wrong results, timing only). Built by
// atdn_vslam_amd/build.py into its own libatdn_microbench.so, which links against libatdn_hip.so: none of these
// template variants is part of the product library.
#include "../../include/atdn_hip.h"
#include "../../atdn_vslam_amd/csrc/conv_sf_dispatch_impl.h"
#include "../../atdn_vslam_amd/csrc/conv_sf6.h"
#include "../../atdn_vslam_amd/csrc/kernels.h"

using namespace atdn;

// us_out[12]: the halo kernel at 8x16 pixels x 256 / 128 / 64 channels and its ablation ladder on the 256-wide block
// (diagnostic builds that skip work: wrong results, timing only); entries 7-11 unused.
extern "C" int atdn_microbench_conv(int nimg, int H, int W, int C, int N, int KH, int KW, int reps, float* us_out) {
  try {
    hipStream_t st = nullptr;
    const long npix = (long)nimg * H * W;
    float *x, *w, *wf, *y, *bias;
    const int K = KH * KW * C;
    ATDN_CHECK(C % 32 == 0 && N % 32 == 0, "microbench shapes are multiples of 32");
    ATDN_HIP(hipMalloc(&x, npix * C * 4)); ATDN_HIP(hipMalloc(&w, (long)N * K * 4)); ATDN_HIP(hipMalloc(&wf, (long)N * K * 4));
    ATDN_HIP(hipMalloc(&y, npix * N * 4)); ATDN_HIP(hipMalloc(&bias, N * 4));
    // pseudo-random f16 bit patterns of moderate magnitude (zero data would overclock the chip); ATDN_MB_ZERO=1
    // shows how much of a time is the clock the chip holds on real operands
    std::vector<unsigned short> hx((size_t)npix * C * 2), hw((size_t)N * K * 2);
    unsigned v = 12345u;
    for (auto& e : hx) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    for (auto& e : hw) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    if (getenv("ATDN_MB_ZERO")) { std::fill(hx.begin(), hx.end(), 0); std::fill(hw.begin(), hw.end(), 0); }
    ATDN_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemcpy(wf, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));  // same bytes: order is irrelevant for timing
    ATDN_HIP(hipMemset(bias, 0, N * 4));
    ConvShape s;
    s.src0 = x; s.ld0 = C; s.sb0 = (long)H * W * C; s.C0 = C; s.H = H; s.W = W;
    s.KH = KH; s.KW = KW; s.stride = 1; s.padH = KH / 2; s.padW = KW / 2;
    s.w = w; s.ldw = K; s.N = N; s.nimg = nimg;
    using E = SfBias<ACT_RELU>;
    E ep{bias, y, (long)H * W * N, N};
    auto time_it = [&](auto&& go) {
      hipEvent_t a, b;
      ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
      for (int i = 0; i < 2 * reps; ++i) go();  // warm-up: clocks and caches settle before the timed launches
      ATDN_HIP(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) go();
      ATDN_HIP(hipEventRecord(b, st));
      ATDN_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, a, b));
      (void)hipEventDestroy(a); (void)hipEventDestroy(b);
      return ms * 1000.f / reps;
    };
    s.wfrag16 = wf;
#define ATDN_MB_SF6(BN, WM, WN, ABL)                                                                              \
    time_it([&]() {                                                                                               \
      if (KH == 3 && KW == 3) launch_conv_sf6<8, BN, WM, WN, 3, 3, E, false, false, ABL>(s, 1.f, ep, st);         \
      else if (KH == 1 && KW == 5) launch_conv_sf6<8, BN, WM, WN, 1, 5, E, false, false, ABL>(s, 1.f, ep, st);    \
      else launch_conv_sf6<8, BN, WM, WN, 5, 1, E, false, false, ABL>(s, 1.f, ep, st);                            \
    })
    us_out[0] = ATDN_MB_SF6(256, 1, 8, 0);
    us_out[1] = ATDN_MB_SF6(128, 1, 4, 0);
    us_out[2] = ATDN_MB_SF6(64, 2, 2, 0);
    us_out[3] = ATDN_MB_SF6(128, 1, 4, 8);    // 128-wide block, no epilogue
    us_out[4] = ATDN_MB_SF6(128, 1, 4, 9);    // ... and no weight loads in the loop
    us_out[5] = ATDN_MB_SF6(128, 1, 4, 13);   // ... and no LDS reads in the loop
    us_out[6] = ATDN_MB_SF6(128, 1, 4, 15);   // ... and no patch refresh: the bare MFMA stream of this tiling
    // (round 4 also timed wave tiles of 64 px x 64 ch — <128, 2, 2> at two waves per SIMD, <256, 2, 4> — against the 128 px x 32 ch
    // ones: half the LDS fragment reads per MFMA for twice the weight loads. Equal within 1 % on every shape:
    // profiles/r04_microbench_conv_tiles.txt. The rows were removed again, the kernel keeps its (threads, 1) launch bounds.)
    us_out[7] = us_out[8] = us_out[9] = 0.f;
#undef ATDN_MB_SF6
    us_out[10] = us_out[11] = 0.f;
    us_out[12] = us_out[13] = us_out[14] = us_out[15] = 0.f;
    if (KH == 3 && KW == 3) {
      // 3x3 only: the tall (12x16 px) 64-wide block and the 96-wide 2x3-wave blocks (N = 192 as two of them)
      us_out[12] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E>(s, 1.f, ep, st); });
      if (N % 96 == 0) {
        us_out[13] = time_it([&]() { launch_conv_sf6<8, 96, 2, 3, 3, 3, E>(s, 1.f, ep, st); });
        us_out[14] = time_it([&]() { launch_conv_sf6<12, 96, 2, 3, 3, 3, E>(s, 1.f, ep, st); });
      }
    }
    if (KH == 1 && KW == 5 && N == 256 && C == 384) {
      // round 3: what the ConvGRU gate epilogue costs — the 128-wide block the pipeline uses, with SfBias and with SfGruZR
      float *hbuf, *zbuf, *rhbuf, *pre;
      ATDN_HIP(hipMalloc(&hbuf, npix * 128 * 4)); ATDN_HIP(hipMalloc(&zbuf, npix * 128 * 4));
      ATDN_HIP(hipMalloc(&rhbuf, npix * 128 * 4)); ATDN_HIP(hipMalloc(&pre, npix * 256 * 4));
      ATDN_HIP(hipMemset(hbuf, 0, npix * 128 * 4)); ATDN_HIP(hipMemset(pre, 0, npix * 256 * 4));
      SfGruZR eg{bias, hbuf, zbuf, rhbuf, (long)H * W * 128, pre, (long)H * W * 256};
      us_out[10] = time_it([&]() { launch_conv_sf6<8, 128, 1, 4, 1, 5, E>(s, 1.f, ep, st); });
      us_out[11] = time_it([&]() { launch_conv_sf6<8, 128, 1, 4, 1, 5, SfGruZR>(s, 1.f, eg, st); });
      (void)hipFree(hbuf); (void)hipFree(zbuf); (void)hipFree(rhbuf); (void)hipFree(pre);
    }
    (void)hipFree(x); (void)hipFree(w); (void)hipFree(wf); (void)hipFree(y); (void)hipFree(bias);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}

// Thin-layer variant of the ladder (encoder shapes: few chunks per tile, so prologue/epilogue weigh more): the
// 64-channel 2x2-wave block at tile heights 8 and 12, its ablations, the single-patch-image variants, and (entries 10, 11)
// the 16x16x32 loop at tile heights 8 and 12. us_out[12].
extern "C" int atdn_microbench_conv_thin(int nimg, int H, int W, int C, int N, int reps, float* us_out) {
  try {
    hipStream_t st = nullptr;
    const long npix = (long)nimg * H * W;
    float *x, *wf, *y, *bias;
    const int K = 9 * C;
    ATDN_CHECK(C % 32 == 0 && N % 32 == 0, "microbench shapes are multiples of 32");
    ATDN_HIP(hipMalloc(&x, npix * C * 4)); ATDN_HIP(hipMalloc(&wf, (long)N * K * 4));
    ATDN_HIP(hipMalloc(&y, npix * N * 4)); ATDN_HIP(hipMalloc(&bias, N * 4));
    std::vector<unsigned short> hx((size_t)npix * C * 2), hw((size_t)N * K * 2);
    unsigned v = 12345u;
    for (auto& e : hx) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    for (auto& e : hw) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    ATDN_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemcpy(wf, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemset(bias, 0, N * 4));
    ConvShape s;
    s.src0 = x; s.ld0 = C; s.sb0 = (long)H * W * C; s.C0 = C; s.H = H; s.W = W;
    s.KH = 3; s.KW = 3; s.stride = 1; s.padH = 1; s.padW = 1;
    s.w = wf; s.ldw = K; s.N = N; s.nimg = nimg;
    using E = SfBias<ACT_RELU>;
    E ep{bias, y, (long)H * W * N, N};
    auto time_it = [&](auto&& go) {
      hipEvent_t a, b;
      ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
      for (int i = 0; i < 2 * reps; ++i) go();
      ATDN_HIP(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) go();
      ATDN_HIP(hipEventRecord(b, st));
      ATDN_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, a, b));
      (void)hipEventDestroy(a); (void)hipEventDestroy(b);
      return ms * 1000.f / reps;
    };
    s.wfrag16 = wf;
    us_out[0] = time_it([&]() { launch_conv_sf6<8, 64, 2, 2, 3, 3, E>(s, 1.f, ep, st); });
    us_out[1] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E>(s, 1.f, ep, st); });
    us_out[2] = time_it([&]() { launch_conv_sf6<16, 64, 4, 2, 3, 3, E>(s, 1.f, ep, st); });
    us_out[3] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 8>(s, 1.f, ep, st); });    // no epilogue
    us_out[4] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 9>(s, 1.f, ep, st); });    // ... no weight loads
    us_out[5] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 13>(s, 1.f, ep, st); });   // ... no LDS reads
    us_out[6] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 15>(s, 1.f, ep, st); });   // bare MFMA stream
    us_out[7] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 1>(s, 1.f, ep, st); });    // only: no weight loads
    us_out[8] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 2>(s, 1.f, ep, st); });    // only: no patch refresh
    us_out[9] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 4>(s, 1.f, ep, st); });    // only: no LDS reads
    us_out[10] = time_it([&]() { launch_conv_sf6<12, 64, 2, 2, 3, 3, E, false, false, 16>(s, 1.f, ep, st); });   // stores stay in L2
    us_out[11] = 0.f;
    (void)hipFree(x); (void)hipFree(wf); (void)hipFree(y); (void)hipFree(bias);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}

// ---- MFMA ceilings under DVFS: the split-f16 product pattern (3 MFMAs per operand pair) on random or zero data,
// operands held in registers or re-read from a conflict-free LDS image every step, for both f16 MFMA shapes.
namespace atdn {
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int SHAPE, int SRC>
__global__ __launch_bounds__(256) void mfma_ceiling_kernel(const float* __restrict__ data, float* __restrict__ out, int steps) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 32];  // A image 128 rows x 128 B, B image likewise
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * 32; i += 256) lds[i] = data[i];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const char* Ab = reinterpret_cast<const char*>(lds);
  const char* Bb = Ab + 128 * 128;
  float total = 0.f;
  if constexpr (SHAPE == 0) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
    auto rd = [&]() {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[t][i] = *reinterpret_cast<const f16x8*>(Ab + ((wm * 2 + i) * 32 + r) * 128 + (((2 * t + h) ^ sw) << 4));
          al[t][i] = *reinterpret_cast<const f16x8*>(Ab + ((wm * 2 + i) * 32 + r) * 128 + (((4 + 2 * t + h) ^ sw) << 4));
          bh[t][i] = *reinterpret_cast<const f16x8*>(Bb + ((wn * 2 + i) * 32 + r) * 128 + (((2 * t + h) ^ sw) << 4));
          bl[t][i] = *reinterpret_cast<const f16x8*>(Bb + ((wn * 2 + i) * 32 + r) * 128 + (((4 + 2 * t + h) ^ sw) << 4));
        }
    };
    rd();
    for (int s = 0; s < steps; ++s) {
      if constexpr (SRC == 1) { asm volatile("" ::: "memory"); rd(); }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t][i], bh[t][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t][i], bl[t][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t][i], bh[t][j], acc[i][j], 0, 0, 0);
          }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) total += acc[i][j][e];
  } else {
    const int r = lane & 15, q = lane >> 4, sw = (r >> 1) & 7;  // 16 rows x 4 k-groups of 8
    f32x4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    f16x8 ah[4], al[4], bh[4], bl[4];
    auto rd = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(Ab + ((wm * 4 + i) * 16 + r) * 128 + ((q ^ sw) << 4));
        al[i] = *reinterpret_cast<const f16x8*>(Ab + ((wm * 4 + i) * 16 + r) * 128 + (((4 + q) ^ sw) << 4));
        bh[i] = *reinterpret_cast<const f16x8*>(Bb + ((wn * 4 + i) * 16 + r) * 128 + ((q ^ sw) << 4));
        bl[i] = *reinterpret_cast<const f16x8*>(Bb + ((wn * 4 + i) * 16 + r) * 128 + (((4 + q) ^ sw) << 4));
      }
    };
    rd();
    for (int s = 0; s < steps; ++s) {
      if constexpr (SRC == 1) { asm volatile("" ::: "memory"); rd(); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) total += acc[i][j][e];
  }
  out[blockIdx.x * 256 + tid] = total;
}

template <int SHAPE, int SRC>
static float time_ceiling(const float* data, float* out, int steps, int launches, hipStream_t st) {
  hipEvent_t a, b;
  ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
  for (int i = 0; i < launches / 4 + 1; ++i) hipLaunchKernelGGL((mfma_ceiling_kernel<SHAPE, SRC>), dim3(512), dim3(256), 0, st, data, out, steps);
  ATDN_HIP(hipEventRecord(a, st));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((mfma_ceiling_kernel<SHAPE, SRC>), dim3(512), dim3(256), 0, st, data, out, steps);
  ATDN_HIP(hipEventRecord(b, st));
  ATDN_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  ATDN_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  const double flop = 512.0 * 4 * steps * (64.0 * 64 * 32 * 2 * 3) * launches;
  return (float)(flop / (ms * 1e-3) / 1e12);
}
}  // namespace atdn

// tf_out[8]: {32x32x16, 16x16x32} x {registers, LDS re-read} x {random, zero} executed TFLOP/s
extern "C" int atdn_microbench_mfma(int steps, int launches, float* tf_out) {
  try {
    hipStream_t st = nullptr;
    float *rnd, *zero, *out;
    const size_t n = 2 * 128 * 32;
    ATDN_HIP(hipMalloc(&rnd, n * 4)); ATDN_HIP(hipMalloc(&zero, n * 4)); ATDN_HIP(hipMalloc(&out, 512 * 256 * 4));
    std::vector<unsigned short> h(n * 2);
    unsigned v = 777u;
    for (auto& e : h) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    ATDN_HIP(hipMemcpy(rnd, h.data(), n * 4, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemset(zero, 0, n * 4));
    int k = 0;
    for (const float* d : {(const float*)rnd, (const float*)zero}) {
      tf_out[k++] = time_ceiling<0, 0>(d, out, steps, launches, st);
      tf_out[k++] = time_ceiling<0, 1>(d, out, steps, launches, st);
      tf_out[k++] = time_ceiling<1, 0>(d, out, steps, launches, st);
      tf_out[k++] = time_ceiling<1, 1>(d, out, steps, launches, st);
    }
    (void)hipFree(rnd); (void)hipFree(zero); (void)hipFree(out);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}
#include "pp_probe.hip"

// ---- conv-like MFMA loop (round 3): the question behind VERDICT r2 #1-iv / #5 — does the 16x16x32 shape buy anything in a loop
// that looks like conv_sf6's (pixel operands re-read from a conflict-free LDS image every step, weight operands in
// registers, 8 waves per block = two per SIMD, 128 pixels x 32 channels per wave, split-f16: three MFMAs per product)?
// SHAPE 0: v_mfma_f32_32x32x16_f16, pixel pitch 144 B (lane = (pixel & 31, k half)); SHAPE 1: v_mfma_f32_16x16x32_f16, pixel
// pitch 160 B (lane = (pixel & 15, k quarter): the pitch that makes those ds_read_b128 conflict-free).
namespace atdn {
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_convlike_kernel(const float* __restrict__ data, float* __restrict__ out, int steps) {
  constexpr int PITCH = SHAPE == 0 ? 144 : 160;
  __shared__ __attribute__((aligned(16))) char lds[2 * 128 * PITCH];   // two images (stand-ins for two taps) of 128 pixels
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 128 * PITCH / 4; i += 512) reinterpret_cast<float*>(lds)[i] = data[i % (2 * 128 * 32)];
  __syncthreads();
  float total = 0.f;
  // weight operands: 32 channels x K = 32, hi and lo
  f16x8 wh[2], wl[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    wh[t] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(data) + ((tid * 2 + t) * 32) % (2 * 128 * 128 - 32));
    wl[t] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(data) + ((tid * 2 + t) * 32 + 16) % (2 * 128 * 128 - 32));
  }
  if constexpr (SHAPE == 0) {
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 ah[2][4], al[2][4];
    auto rd = [&](int set, int img, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* p = lds + img * 128 * PITCH + (i * 32 + r) * PITCH + 32 * t + 16 * h;
        ah[set][i] = *reinterpret_cast<const f16x8*>(p);
        al[set][i] = *reinterpret_cast<const f16x8*>(p + 64);
      }
    };
    rd(0, 0, 0);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {     // two taps x two half-steps
        __builtin_amdgcn_sched_barrier(0);
        rd((hs + 1) & 1, ((hs + 1) >> 1) & 1, (hs + 1) & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[hs & 1], al[hs & 1][i], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[hs & 1], ah[hs & 1][i], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[hs & 1], ah[hs & 1][i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (k < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) total += acc[i][e];
  } else {
    const int r = lane & 15, q = lane >> 4;
    f32x4_t acc[8][2];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    f16x8 ah[2][4], al[2][4];
    auto rd = [&](int set, int img, int half) __attribute__((always_inline)) {   // pixel blocks 4 half .. 4 half + 3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* p = lds + img * 128 * PITCH + ((half * 4 + i) * 16 + r) * PITCH + 16 * q;
        ah[set][i] = *reinterpret_cast<const f16x8*>(p);
        al[set][i] = *reinterpret_cast<const f16x8*>(p + 64);
      }
    };
    rd(0, 0, 0);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {     // two taps x two halves of the wave's pixel blocks
        __builtin_amdgcn_sched_barrier(0);
        rd((hs + 1) & 1, ((hs + 1) >> 1) & 1, (hs + 1) & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[(hs & 1) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al[hs & 1][i], acc[(hs & 1) * 4 + i][j], 0, 0, 0);
            acc[(hs & 1) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah[hs & 1][i], acc[(hs & 1) * 4 + i][j], 0, 0, 0);
            acc[(hs & 1) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[hs & 1][i], acc[(hs & 1) * 4 + i][j], 0, 0, 0);
          }
#pragma unroll
        for (int k = 0; k < 24; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (k < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) total += acc[i][j][e];
  }
  out[blockIdx.x * 512 + tid] = total;
}

template <int SHAPE>
static float time_convlike(const float* data, float* out, int steps, int launches, hipStream_t st) {
  hipEvent_t a, b;
  ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
  for (int i = 0; i < launches / 4 + 1; ++i) hipLaunchKernelGGL((mfma_convlike_kernel<SHAPE>), dim3(256), dim3(512), 0, st, data, out, steps);
  ATDN_HIP(hipEventRecord(a, st));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((mfma_convlike_kernel<SHAPE>), dim3(256), dim3(512), 0, st, data, out, steps);
  ATDN_HIP(hipEventRecord(b, st));
  ATDN_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  ATDN_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  // per step and wave: 2 taps x (128 px x 32 ch x K 32) x 3 products
  const double flop = 256.0 * 8 * steps * (2.0 * 128 * 32 * 32 * 2 * 3) * launches;
  return (float)(flop / (ms * 1e-3) / 1e12);
}
}  // namespace atdn

// tf_out[4]: {32x32x16, 16x16x32} x {random, zero} executed TFLOP/s of the conv-like loop
extern "C" int atdn_microbench_mfma_convlike(int steps, int launches, float* tf_out) {
  try {
    hipStream_t st = nullptr;
    float *rnd, *zero, *out;
    const size_t n = 2 * 128 * 32;
    ATDN_HIP(hipMalloc(&rnd, n * 4)); ATDN_HIP(hipMalloc(&zero, n * 4)); ATDN_HIP(hipMalloc(&out, 256 * 512 * 4));
    std::vector<unsigned short> h(n * 2);
    unsigned v = 777u;
    for (auto& e : h) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    ATDN_HIP(hipMemcpy(rnd, h.data(), n * 4, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemset(zero, 0, n * 4));
    int k = 0;
    for (const float* d : {(const float*)rnd, (const float*)zero}) {
      tf_out[k++] = time_convlike<0>(d, out, steps, launches, st);
      tf_out[k++] = time_convlike<1>(d, out, steps, launches, st);
    }
    (void)hipFree(rnd); (void)hipFree(zero); (void)hipFree(out);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}

// ---- HBM stream rates (round 4): what a kernel that WRITES a few GB can expect. The correlation volume (3.56 GB at 16 pairs)
// and the attention matrix (2.5 GB) are written once per forward by MFMA-heavy kernels whose stores looked expensive; this
// prices the stores alone: 16 bytes per lane, every CU, `bytes` per launch (> 256 MiB: past the Infinity Cache).
// gb_out[7] (GB/s): plain store, non-temporal store, read (sum kept), copy (read + write bytes counted), store in 4 KB runs
// scattered over the buffer (one wave writes 4 KB, the next wave's run is 30 KB further), the same with 128-byte runs, and
// 64-byte half-line stores (16 per instruction, the other halves by the wave's next instruction).
namespace atdn {
typedef float v4s __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(float* __restrict__ buf, const float* __restrict__ src, long n16, float* __restrict__ sink) {
  const long stride = (long)gridDim.x * 256;
  v4s acc = {0.f, 0.f, 0.f, 0.f};
  const v4s val = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    if (MODE == 0) reinterpret_cast<v4s*>(buf)[i] = val;
    else if (MODE == 1) __builtin_nontemporal_store(val, reinterpret_cast<v4s*>(buf) + i);
    else if (MODE == 2) acc += reinterpret_cast<const v4s*>(src)[i];
    else if (MODE == 3) reinterpret_cast<v4s*>(buf)[i] = reinterpret_cast<const v4s*>(src)[i];
    else if (MODE == 6) {
      // half-line stores as an MFMA accumulator leaves them: one wave instruction = 16 segments of 64 B (4 lanes each), 128 B
      // apart; the NEXT instruction of the wave fills the other halves of the same 16 lines
      const long w = i >> 6;            // wave-instruction index
      const int l = (int)(i & 63);
      const long dst = (w >> 1) * 2048 + (l >> 2) * 128 + (w & 1) * 64 + (l & 3) * 16;
      __builtin_nontemporal_store(val, reinterpret_cast<v4s*>(reinterpret_cast<char*>(buf) + dst));
    } else {
      // scattered runs: wave-instruction w (64 lanes x 16 B = 1 KB) belongs to run w / RUNK of RUN bytes; consecutive runs are
      // ROWB bytes apart (a correlation row of 7680 floats), wrapping over the buffer
      constexpr long RUN = MODE == 4 ? 4096 : 128, ROWB = 30720;
      const long byte = i * 16;
      const long run = byte / RUN, within = byte - run * RUN;
      const long nrows = (n16 * 16) / ROWB;
      const long row = run % nrows, col = (run / nrows) * RUN;
      const long dst = row * ROWB + (col % ROWB) + within;
      __builtin_nontemporal_store(val, reinterpret_cast<v4s*>(reinterpret_cast<char*>(buf) + (dst & ~15L)));
    }
  }
  if (MODE == 2 && acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;
}
}  // namespace atdn

extern "C" int atdn_microbench_stream(long bytes, int reps, float* gb_out) {
  try {
    hipStream_t st = nullptr;
    float *a, *b, *sink;
    ATDN_HIP(hipMalloc(&a, bytes)); ATDN_HIP(hipMalloc(&b, bytes)); ATDN_HIP(hipMalloc(&sink, 64));
    ATDN_HIP(hipMemset(a, 0, bytes)); ATDN_HIP(hipMemset(b, 0, bytes));
    const long n16 = bytes / 16;
    auto run = [&](auto kern, double moved) {
      hipEvent_t e0, e1;
      ATDN_HIP(hipEventCreate(&e0)); ATDN_HIP(hipEventCreate(&e1));
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, st, a, b, n16, sink);
      ATDN_HIP(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, st, a, b, n16, sink);
      ATDN_HIP(hipEventRecord(e1, st));
      ATDN_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, e0, e1));
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      return (float)(moved * reps / (ms * 1e-3) / 1e9);
    };
    gb_out[0] = run(atdn::stream_kernel<0>, (double)bytes);
    gb_out[1] = run(atdn::stream_kernel<1>, (double)bytes);
    gb_out[2] = run(atdn::stream_kernel<2>, (double)bytes);
    gb_out[3] = run(atdn::stream_kernel<3>, 2.0 * bytes);
    gb_out[4] = run(atdn::stream_kernel<4>, (double)bytes);
    gb_out[5] = run(atdn::stream_kernel<5>, (double)bytes);
    gb_out[6] = run(atdn::stream_kernel<6>, (double)bytes);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}

// ---- strip streams (round 4): the read pattern of attention x V. Every wave of a 512-thread block (two blocks per CU) streams ITS
// OWN contiguous run of `run_bytes` in 3 KB steps (2 x 1 KB as 16 B per lane + 2 x 512 B as 8 B per lane), three steps ahead,
// straight into registers — 3,712 concurrent sequential streams 0.7 MB apart. The plain read test above has all waves marching
// through the buffer together. Modes: 0 as the kernel does it (non-temporal); 1 the eight runs of a block interleaved per step
// (a block reads 24 KB contiguous per step); 2 default cache policy; 3 the step as 3 x 1 KB of 16-B loads; 4 mode 0 with two
// block barriers per step (the ping-pong's coupling); 5 mode 1 with the barriers; 6 mode 0, ring six steps deep.
namespace atdn {
typedef unsigned int u2s __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void strip_stream_kernel(const char* __restrict__ buf, long run_bytes, int nstep, float* __restrict__ sink) {
  constexpr int D = MODE == 6 ? 6 : 3;
  constexpr bool INTER = MODE == 1 || MODE == 5, NT = MODE != 2, BAR = MODE == 4 || MODE == 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = buf + (long)blockIdx.x * 8 * run_bytes + (INTER ? (long)wave * 3072 : (long)wave * run_bytes);
  const long step = INTER ? 8L * 3072 : 3072L;
  v4s r16[D][3];
  u2s r8[D][2];
  auto load = [&](int q, int slot) __attribute__((always_inline)) {
    const char* p = base + (long)min(q, nstep - 1) * step;
    if (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 3; ++k) r16[slot][k] = __builtin_nontemporal_load(reinterpret_cast<const v4s*>(p + k * 1024 + lane * 16));
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        r16[slot][k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4s*>(p + k * 1024 + lane * 16))
                          : *reinterpret_cast<const v4s*>(p + k * 1024 + lane * 16);
        r8[slot][k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u2s*>(p + 2048 + k * 512 + lane * 8))
                         : *reinterpret_cast<const u2s*>(p + 2048 + k * 512 + lane * 8);
      }
    }
  };
  v4s acc = {0.f, 0.f, 0.f, 0.f};
  unsigned acci = 0;
#pragma unroll
  for (int c = 0; c < D; ++c) load(c, c);
  for (int q0 = 0; q0 < nstep; q0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (MODE == 3) acc += r16[d][0] + r16[d][1] + r16[d][2];
      else { acc += r16[d][0] + r16[d][1]; acci += r8[d][0][0] ^ r8[d][0][1] ^ r8[d][1][0] ^ r8[d][1][1]; }
      load(q0 + d + D, d);
      if (BAR) { __syncthreads(); __syncthreads(); }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w + (float)acci == 1.2345e-30f) sink[0] = acc.x;
}
}  // namespace atdn

// gb_out[7]: GB/s of modes 0..6 over nblocks x 8 runs of run_bytes (rounded down to whole 3 KB steps, a multiple of 6)
extern "C" int atdn_microbench_strips(long run_bytes, int nblocks, int reps, float* gb_out) {
  try {
    hipStream_t st = nullptr;
    const int nstep = (int)(run_bytes / 3072) / 6 * 6;
    run_bytes = (long)nstep * 3072;
    const long bytes = run_bytes * 8 * nblocks;
    char* a; float* sink;
    ATDN_HIP(hipMalloc(&a, bytes)); ATDN_HIP(hipMalloc(&sink, 64));
    ATDN_HIP(hipMemset(a, 0, bytes));
    auto run = [&](auto kern) {
      hipEvent_t e0, e1;
      ATDN_HIP(hipEventCreate(&e0)); ATDN_HIP(hipEventCreate(&e1));
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), 0, st, a, run_bytes, nstep, sink);
      ATDN_HIP(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), 0, st, a, run_bytes, nstep, sink);
      ATDN_HIP(hipEventRecord(e1, st));
      ATDN_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, e0, e1));
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      return (float)((double)bytes * reps / (ms * 1e-3) / 1e9);
    };
    gb_out[0] = run(atdn::strip_stream_kernel<0>);
    gb_out[1] = run(atdn::strip_stream_kernel<1>);
    gb_out[2] = run(atdn::strip_stream_kernel<2>);
    gb_out[3] = run(atdn::strip_stream_kernel<3>);
    gb_out[4] = run(atdn::strip_stream_kernel<4>);
    gb_out[5] = run(atdn::strip_stream_kernel<5>);
    gb_out[6] = run(atdn::strip_stream_kernel<6>);
    (void)hipFree(a); (void)hipFree(sink);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}
