// Frame front-end of the odometry path: what NeuralSLAM does to every camera frame before the flow
// network sees it — `im.to(device)`, `TF.resize(im, (376, 1232))`, `InputPadder.pad`
// (atdn_vslam/slam_framework/neural_slam.py:197-199,219-221; whl:GMA/core/utils/utils.py:8-20).
//
// * resize: ONE kernel for both axes, source uint8 or fp32. ATen's separable kernels interpolate the last dimension
//   first and round that intermediate to fp32; here every output pixel recomputes the (few) horizontal sums it needs
//   in the same order, so results are bit-identical to the two-pass form and no intermediate buffer exists (the
//   round-1 version kept a process-global one, shared by every stream).
//   antialias = 1: F.interpolate(bilinear, antialias=True, align_corners=False) = torchvision >= 0.17 tensor resize;
//   antialias = 0: F.interpolate(bilinear, align_corners=False) = what older torchvision does for tensors.
// * replicate padding to multiples of 8 as its own small kernel (no ATen op left on the boundary).
// * FrameIngest: uint8 frames in (pinned) host memory -> async H2D on a private copy stream into one of two device
//   staging slots -> resize (+ uint8 -> fp32) on the caller's stream. The copy of clip k+1 overlaps the flow network
//   of clip k; events order slot reuse.
#include "frontend.h"

#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace atdn {

namespace {

std::vector<ResizeTable> make_table_aa(int in, int out) {
  // ATen upsample_bilinear2d_aa weight computation (HelperInterpLinear::compute_indices_weights_aa), in fp32
  std::vector<ResizeTable> tab(out);
  const float scale = (float)in / (float)out;                  // area_pixel_compute_scale, align_corners = false
  const float support = (scale >= 1.0f) ? 1.0f * scale : 1.0f; // interp_size (2) * 0.5 [* scale]
  const float invscale = (scale >= 1.0f) ? 1.0f / scale : 1.0f;
  for (int i = 0; i < out; ++i) {
    const float center = scale * ((float)i + 0.5f);
    int xmin = (int)(center - support + 0.5f); if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5f); if (xmax > in) xmax = in;
    const int size = xmax - xmin;
    ATDN_CHECK(size >= 1 && size <= RESIZE_TAPS, "resize ratio outside the supported range (down-scaling by more than 3.5x)");
    float total = 0.f;
    ResizeTable t{};
    t.start = xmin; t.count = size;
    for (int j = 0; j < size; ++j) {
      float x = ((float)(j + xmin) - center + 0.5f) * invscale;
      if (x < 0.f) x = -x;
      const float w = (x < 1.0f) ? 1.0f - x : 0.0f;
      t.w[j] = w; total += w;
    }
    if (total != 0.f) for (int j = 0; j < size; ++j) t.w[j] /= total;
    tab[i] = t;
  }
  return tab;
}

std::vector<ResizeTable> make_table_bilinear(int in, int out) {
  // ATen upsample_bilinear2d, align_corners = false: src = scale * (dst + 0.5) - 0.5, clamped at 0; taps x0, x0 + 1
  // (the second tap collapses onto the first at the right border)
  std::vector<ResizeTable> tab(out);
  const float scale = (float)in / (float)out;
  for (int i = 0; i < out; ++i) {
    // one rounding (fused multiply-add), as the ATen CPU kernels of this torch build compute it: lambda carries the
    // rounding of `src` (ulp 6e-5 at x ~ 600) times the local contrast into the output, so the contraction matters
    float src = std::fmaf(scale, (float)i + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
    int x0 = (int)src;
    if (x0 > in - 1) x0 = in - 1;
    const float l1 = src - (float)x0, l0 = 1.0f - l1;
    ResizeTable t{};
    t.start = x0;
    if (x0 + 1 <= in - 1) { t.count = 2; t.w[0] = l0; t.w[1] = l1; }
    else { t.count = 1; t.w[0] = 1.0f; }
    tab[i] = t;
  }
  return tab;
}

std::vector<ResizeTable> make_table(int in, int out, int antialias) {
  if (in == out) {   // ATen returns the input unchanged when sizes match
    std::vector<ResizeTable> tab(out);
    for (int i = 0; i < out; ++i) { tab[i] = ResizeTable{}; tab[i].start = i; tab[i].count = 1; tab[i].w[0] = 1.0f; }
    return tab;
  }
  return antialias ? make_table_aa(in, out) : make_table_bilinear(in, out);
}

struct Plan { int dev, Hin, Win, Hout, Wout, aa; ResizeTable* ty; ResizeTable* tx; };
std::mutex g_plans_mutex;
std::vector<Plan> g_plans;

}  // namespace

const ResizePlanRef resize_plan(int Hin, int Win, int Hout, int Wout, int antialias) {
  int dev = 0;
  ATDN_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_plans_mutex);
  for (auto& p : g_plans)
    if (p.dev == dev && p.Hin == Hin && p.Win == Win && p.Hout == Hout && p.Wout == Wout && p.aa == antialias)
      return {p.ty, p.tx};
  Plan p{dev, Hin, Win, Hout, Wout, antialias, nullptr, nullptr};
  const auto ty = make_table(Hin, Hout, antialias), tx = make_table(Win, Wout, antialias);
  ATDN_HIP(hipMalloc(&p.ty, ty.size() * sizeof(ResizeTable)));
  ATDN_HIP(hipMalloc(&p.tx, tx.size() * sizeof(ResizeTable)));
  ATDN_HIP(hipMemcpy(p.ty, ty.data(), ty.size() * sizeof(ResizeTable), hipMemcpyHostToDevice));
  ATDN_HIP(hipMemcpy(p.tx, tx.data(), tx.size() * sizeof(ResizeTable), hipMemcpyHostToDevice));
  g_plans.push_back(p);
  return {p.ty, p.tx};
}

// thread = one output pixel (x fastest: coalesced stores, neighbouring lanes share source lines)
template <class T, bool IDY>
__global__ __launch_bounds__(256) void resize_kernel(const T* __restrict__ src, const ResizeTable* __restrict__ ty,
                                                     const ResizeTable* __restrict__ tx, int planes, int Hin, int Win,
                                                     int Hout, int Wout, float* __restrict__ dst) {
  const long total = (long)planes * Hout * Wout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wout);
    const long r = i / Wout;
    const int y = (int)(r % Hout);
    const long pl = r / Hout;
    const ResizeTable hx = tx[x];
    const T* s = src + pl * (long)Hin * Win + hx.start;
    if (IDY) {   // rows map one to one (376x1241 -> 376x1232): the horizontal sum is the result
      const T* row = s + (long)y * Win;
      float acc = 0.f;
      for (int k = 0; k < hx.count; ++k) acc += hx.w[k] * (float)row[k];
      dst[i] = acc;
    } else {
      const ResizeTable vy = ty[y];
      float out = 0.f;
      for (int j = 0; j < vy.count; ++j) {
        const T* row = s + (long)(vy.start + j) * Win;
        float acc = 0.f;   // = the fp32 intermediate ATen's horizontal pass would have stored
        for (int k = 0; k < hx.count; ++k) acc += hx.w[k] * (float)row[k];
        out += vy.w[j] * acc;
      }
      dst[i] = out;
    }
  }
}

template <class T>
void launch_resize(const T* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst,
                   hipStream_t st) {
  const ResizePlanRef p = resize_plan(Hin, Win, Hout, Wout, antialias);
  const long n = (long)planes * Hout * Wout;
  const dim3 grid((unsigned)std::min<long>(cdivl(n, 256), 16384));
  if (Hin == Hout) hipLaunchKernelGGL((resize_kernel<T, true>), grid, dim3(256), 0, st, src, p.ty, p.tx, planes, Hin, Win, Hout, Wout, dst);
  else hipLaunchKernelGGL((resize_kernel<T, false>), grid, dim3(256), 0, st, src, p.ty, p.tx, planes, Hin, Win, Hout, Wout, dst);
  ATDN_HIP(hipGetLastError());
}
template void launch_resize<float>(const float*, int, int, int, int, int, int, float*, hipStream_t);
template void launch_resize<unsigned char>(const unsigned char*, int, int, int, int, int, int, float*, hipStream_t);

// F.pad(x, [l, r, t, b], mode="replicate") on [planes, H, W]
__global__ __launch_bounds__(256) void pad_replicate_kernel(const float* __restrict__ src, int planes, int H, int W, int l,
                                                            int t, int Ho, int Wo, float* __restrict__ dst) {
  const long total = (long)planes * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wo);
    const long r = i / Wo;
    const int y = (int)(r % Ho);
    const long pl = r / Ho;
    const int sx = min(max(x - l, 0), W - 1), sy = min(max(y - t, 0), H - 1);
    dst[i] = src[pl * (long)H * W + (long)sy * W + sx];
  }
}
void launch_pad_replicate(const float* src, int planes, int H, int W, int l, int r, int t, int b, float* dst, hipStream_t st) {
  const int Ho = H + t + b, Wo = W + l + r;
  const long n = (long)planes * Ho * Wo;
  hipLaunchKernelGGL(pad_replicate_kernel, dim3((unsigned)std::min<long>(cdivl(n, 256), 16384)), dim3(256), 0, st, src,
                     planes, H, W, l, t, Ho, Wo, dst);
  ATDN_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ host uint8 frames -> device fp32 frames
FrameIngest::FrameIngest(int Hin_, int Win_, int Hout_, int Wout_, int max_frames_, int antialias_)
    : Hin(Hin_), Win(Win_), Hout(Hout_), Wout(Wout_), max_frames(max_frames_), antialias(antialias_) {
  ATDN_CHECK(Hin >= 1 && Win >= 1 && Hout >= 1 && Wout >= 1 && max_frames >= 1, "bad frame geometry");
  ATDN_HIP(hipGetDevice(&dev_));
  (void)resize_plan(Hin, Win, Hout, Wout, antialias);   // builds the tables (and validates the ratio) now
  const size_t bytes = (size_t)max_frames * 3 * Hin * Win;
  for (int s = 0; s < 2; ++s) {
    ATDN_HIP(hipMalloc(&stage_[s], bytes));
    ATDN_HIP(hipEventCreateWithFlags(&copied_[s], hipEventDisableTiming));
    ATDN_HIP(hipEventCreateWithFlags(&consumed_[s], hipEventDisableTiming));
  }
  ATDN_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
}

FrameIngest::~FrameIngest() {
  DeviceGuard dg(dev_);
  (void)hipDeviceSynchronize();
  for (int s = 0; s < 2; ++s) {
    if (stage_[s]) (void)hipFree(stage_[s]);
    if (copied_[s]) (void)hipEventDestroy(copied_[s]);
    if (consumed_[s]) (void)hipEventDestroy(consumed_[s]);
  }
  if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
}

void FrameIngest::ingest(const unsigned char* host_frames, int n, float* dst, hipStream_t st) {
  ATDN_CHECK(host_frames && dst && n >= 1 && n <= max_frames, "frame count exceeds max_frames of this handle");
  const int s = next_;
  next_ ^= 1;
  // Host-buffer lifetime: the copy issued two calls ago (the previous user of this slot) has finished before this call
  // returns, so a caller only has to keep the host buffers of its last TWO calls alive (include/atdn_hip.h).
  if (used_[s]) ATDN_HIP(hipEventSynchronize(copied_[s]));
  // the slot's previous contents must have been read by the resize kernel that used them
  if (used_[s]) ATDN_HIP(hipStreamWaitEvent(copy_stream_, consumed_[s], 0));
  ATDN_HIP(hipMemcpyAsync(stage_[s], host_frames, (size_t)n * 3 * Hin * Win, hipMemcpyHostToDevice, copy_stream_));
  ATDN_HIP(hipEventRecord(copied_[s], copy_stream_));
  ATDN_HIP(hipStreamWaitEvent(st, copied_[s], 0));
  launch_resize<unsigned char>(stage_[s], n * 3, Hin, Win, Hout, Wout, antialias, dst, st);
  ATDN_HIP(hipEventRecord(consumed_[s], st));
  used_[s] = true;
}

}  // namespace atdn
