// Diagnostic: H3 encode / decode round trip on the device (hipcc --offload-arch=gfx950 tools/diag/h3_check.hip -o gpurun_out/h3_check)
#include "../../atdn_vslam_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace atdn {
void set_last_error(const std::string&) {}
void sf_counter_register(void (*)(unsigned int*)) {}
namespace {
__global__ void rt_kernel(const float* v, float* out_hi, float* out_lo, unsigned* out_b, int n8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  f16x8 hi; u32x2 b;
  bool clamped = false;
  float gs = 0.f;
  for (int k = 0; k < 8; ++k) gs += v[8 * i + k];
  h3_encode(v + 8 * i, gs, hi, b, clamped);
  const f16x8 lo = h3_decode_lo(hi, b);
  for (int k = 0; k < 8; ++k) { out_hi[8 * i + k] = (float)hi[k]; out_lo[8 * i + k] = (float)lo[k]; }
  out_b[2 * i] = b[0]; out_b[2 * i + 1] = b[1];
}
}
}
int main() {
  const int n8 = 4096;
  std::vector<float> v(8 * n8), hi(8 * n8), lo(8 * n8);
  std::vector<unsigned> bb(2 * n8);
  unsigned s = 12345;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; const float u = (s >> 8) / 16777216.0f; s = s * 1664525u + 1013904223u; const float g = (s >> 8) / 16777216.0f; x = 1024.f * expf(-12.f * u) * (0.5f + g); }
  float *dv, *dh, *dl; unsigned* db;
  hipMalloc(&dv, v.size() * 4); hipMalloc(&dh, v.size() * 4); hipMalloc(&dl, v.size() * 4); hipMalloc(&db, bb.size() * 4);
  hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(atdn::rt_kernel, dim3((n8 + 63) / 64), dim3(64), 0, 0, dv, dh, dl, db, n8);
  hipMemcpy(hi.data(), dh, v.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(lo.data(), dl, v.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(bb.data(), db, bb.size() * 4, hipMemcpyDeviceToHost);
  double worst_hi = 0, worst = 0;
  for (int i = 0; i < n8; ++i) {
    float mx = 0; for (int k = 0; k < 8; ++k) mx = fmaxf(mx, hi[8 * i + k]);
    for (int k = 0; k < 8; ++k) {
      worst_hi = fmax(worst_hi, fabs(hi[8 * i + k] - v[8 * i + k]) / mx);
      worst = fmax(worst, fabs(hi[8 * i + k] + lo[8 * i + k] - v[8 * i + k]) / mx);
    }
  }
  printf("hi-only err / group max %.3e ; hi+lo err / group max %.3e\n", worst_hi, worst);
  // the decode against its definition, bit for bit: lo = f16((byte - 128) * 2^(E - 33)), E = f16 exponent (>= 1) of the group's largest hi
  long bad = 0;
  for (int i = 0; i < n8; ++i) {
    float mx = 0; for (int k = 0; k < 8; ++k) mx = fmaxf(mx, hi[8 * i + k]);
    int e = 0; (void)frexpf(mx, &e);                      // mx = f * 2^e, f in [0.5, 1): f16 exponent field = e + 14
    const int E = mx > 0.f ? (e + 14 < 1 ? 1 : e + 14) : 1;
    for (int k = 0; k < 8; ++k) {
      const int byte = (bb[2 * i + (k >> 2)] >> (8 * (k & 3))) & 255;
      const float ref = (float)(_Float16)ldexpf((float)(byte - 128), E - 33);
      if (ref != lo[8 * i + k]) ++bad;
    }
  }
  printf("decode vs definition: %ld mismatches of %d\n", bad, 8 * n8);
  for (int i = 0; i < 2; ++i) { printf("group %d:", i); for (int k = 0; k < 8; ++k) printf(" v %.6g hi %.6g lo %.3g |", v[8*i+k], hi[8*i+k], lo[8*i+k]); printf(" bytes %08x %08x\n", bb[2*i], bb[2*i+1]); }
  return 0;
}
