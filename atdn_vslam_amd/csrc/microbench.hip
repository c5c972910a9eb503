// Diagnostic micro-benchmark of the halo-patch kernels on a ConvGRU-shaped problem (not on the product path).
#include "../../include/atdn_hip.h"
#include "conv_sf_dispatch_impl.h"
#include "kernels.h"

namespace atdn {
template <int ABL>
static float time_sf3(const ConvShape& s, const SfBias<ACT_RELU>& ep, int reps, hipStream_t st) {
  hipEvent_t a, b;
  ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
  launch_conv_sf3<2, SfBias<ACT_RELU>, ABL>(s, 1.f, ep, st);
  ATDN_HIP(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) launch_conv_sf3<2, SfBias<ACT_RELU>, ABL>(s, 1.f, ep, st);
  ATDN_HIP(hipEventRecord(b, st));
  ATDN_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  ATDN_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms * 1000.f / reps;
}
}  // namespace atdn
using namespace atdn;

extern "C" int atdn_microbench_conv(int nimg, int H, int W, int C, int N, int KH, int KW, int reps, float* us_out) {
  try {
    hipStream_t st = nullptr;
    const long npix = (long)nimg * H * W;
    float *x, *w, *y, *bias;
    const int K = KH * KW * C;
    ATDN_HIP(hipMalloc(&x, npix * C * 4)); ATDN_HIP(hipMalloc(&w, (long)N * K * 4));
    ATDN_HIP(hipMalloc(&y, npix * N * 4)); ATDN_HIP(hipMalloc(&bias, N * 4));
    // pseudo-random f16 bit patterns of moderate magnitude (zero data would overclock the chip)
    std::vector<unsigned short> hx((size_t)npix * C * 2), hw((size_t)N * K * 2);
    unsigned v = 12345u;
    for (auto& e : hx) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    for (auto& e : hw) { v = v * 1664525u + 1013904223u; e = (unsigned short)(0x3000 + ((v >> 16) & 0x0FFF) + ((v >> 31) << 15)); }
    ATDN_HIP(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    ATDN_HIP(hipMemset(bias, 0, N * 4));
    ConvShape s;
    s.src0 = x; s.ld0 = C; s.sb0 = (long)H * W * C; s.C0 = C; s.H = H; s.W = W;
    s.KH = KH; s.KW = KW; s.stride = 1; s.padH = KH / 2; s.padW = KW / 2;
    s.w = w; s.ldw = K; s.N = N; s.nimg = nimg;
    SfBias<ACT_RELU> ep{bias, y, (long)H * W * N, N};
    us_out[0] = time_sf3<0>(s, ep, reps, st);
    us_out[1] = time_sf3<1>(s, ep, reps, st);
    us_out[2] = time_sf3<3>(s, ep, reps, st);
    us_out[3] = time_sf3<7>(s, ep, reps, st);
    {  // generation 4, 16x16 tiles
      hipEvent_t a, b;
      ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
      launch_conv_sf4<2, SfBias<ACT_RELU>, 16>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) launch_conv_sf4<2, SfBias<ACT_RELU>, 16>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(b, st));
      ATDN_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, a, b));
      us_out[4] = ms * 1000.f / reps;
    }
    {  // generation 4: weight tiles by LDS-DMA, 8x16 tiles
      hipEvent_t a, b;
      ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
      launch_conv_sf4<2, SfBias<ACT_RELU>, 8>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) launch_conv_sf4<2, SfBias<ACT_RELU>, 8>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(b, st));
      ATDN_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, a, b));
      us_out[5] = ms * 1000.f / reps;
    }
    {  // generation 2 with 16x16 output tiles (512 threads)
      hipEvent_t a, b;
      ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
      launch_conv_sf2<2, SfBias<ACT_RELU>, 16>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) launch_conv_sf2<2, SfBias<ACT_RELU>, 16>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(b, st));
      ATDN_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, a, b));
      us_out[6] = ms * 1000.f / reps;
    }
    {  // generation 2 (single-role waves) for reference
      hipEvent_t a, b;
      ATDN_HIP(hipEventCreate(&a)); ATDN_HIP(hipEventCreate(&b));
      launch_conv_sf2<2>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(a, st));
      for (int i = 0; i < reps; ++i) launch_conv_sf2<2>(s, 1.f, ep, st);
      ATDN_HIP(hipEventRecord(b, st));
      ATDN_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      ATDN_HIP(hipEventElapsedTime(&ms, a, b));
      us_out[7] = ms * 1000.f / reps;
    }
    (void)hipFree(x); (void)hipFree(w); (void)hipFree(y); (void)hipFree(bias);
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}
