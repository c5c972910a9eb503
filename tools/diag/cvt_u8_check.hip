// Diagnostic: rounding and saturation of v_cvt_pk_u8_f32 (hipcc --offload-arch=gfx950 tools/diag/cvt_u8_check.hip -o tools/diag/cvt_u8_check.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* y, int n) {
  const int i = threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main() {
  const float h[] = {0.5f, 1.5f, 2.5f, 2.4f, 2.6f, 3.5f, 254.5f, 255.5f, 256.7f, 1000.f, -0.3f, -0.5f, -0.6f, 127.5f, 128.5f, 0.49999997f};
  const int n = sizeof(h) / sizeof(float);
  float* dx; unsigned* dy; unsigned out[32];
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, sizeof(out));
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, n);
  hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], out[i]);
  return 0;
}
