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

// thread = one output pixel (x fastest: coalesced stores, neighbouring lanes share source lines); block row = one output row of
// one plane (blockIdx.y), so the row / plane split is scalar arithmetic once per block — the flat index of rounds 2-3 cost every
// pixel two 64-bit divisions (125 us per 17-frame clip for 118 MB of traffic)
template <class T, bool IDY>
__global__ __launch_bounds__(256) void resize_kernel(const T* __restrict__ src, const ResizeTable* __restrict__ ty,
                                                     const ResizeTable* __restrict__ tx, int planes, int Hin, int Win,
                                                     int Hout, int Wout, float* __restrict__ dst) {
  const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (x >= Wout) return;
  const int r = (int)blockIdx.y;           // plane * Hout + y
  const int pl = r / Hout, y = r - pl * Hout;
  const long i = (long)r * Wout + x;
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

// Rows map one to one (376 x 1241 -> 376 x 1232, the KITTI clip of every benchmark step): a block = 256 output columns of
// RZ_ROWS rows of one plane. Each thread keeps ITS column's table entry in registers for all rows, and the block stages the
// ~265 source elements its columns touch per row in LDS with one coalesced load per element — the per-pixel form above issues
// a table load, a byte load and a weight load per TAP (8 vector memory instructions per pixel; 116 us per 17-frame clip for
// 118 MB of traffic). Same products in the same order: bit-identical.
constexpr int RZ_ROWS = 8, RZ_SPAN = 256 + 256 / 4 + 2 * RESIZE_TAPS;   // source span of 256 columns (ratios up to 1.25) + taps
template <class T>
__global__ __launch_bounds__(256) void resize_rows_kernel(const T* __restrict__ src, const ResizeTable* __restrict__ tx, int H,
                                                          int Win, int Wout, float* __restrict__ dst) {
  __shared__ T srow[RZ_ROWS][RZ_SPAN];
  const int tid = threadIdx.x, x0 = (int)blockIdx.x * 256, x = x0 + tid;
  const int y0 = (int)blockIdx.y * RZ_ROWS, pl = (int)blockIdx.z;
  const int xl = min(x0 + 255, Wout - 1);
  const int smin = tx[x0].start, smax = tx[xl].start + tx[xl].count;   // (starts are non-decreasing in x)
  const int span = min(smax - smin, RZ_SPAN);
  const T* sp = src + ((long)pl * H + y0) * Win + smin;
  const int nrow = min(RZ_ROWS, H - y0);
  for (int r = 0; r < nrow; ++r)
    for (int i = tid; i < span; i += 256) srow[r][i] = sp[(long)r * Win + i];
  int start = 0, count = 0;
  float w[RESIZE_TAPS];
#pragma unroll
  for (int k = 0; k < RESIZE_TAPS; ++k) w[k] = 0.f;
  if (x < Wout) {
    const ResizeTable* t = tx + x;
    start = t->start - smin; count = t->count;
    // (a table entry is 40 bytes with the weights at offset 8: 8-byte aligned, so four float2 loads, not two float4)
#pragma unroll
    for (int k = 0; k < RESIZE_TAPS; k += 2) {
      const float2 wk = *reinterpret_cast<const float2*>(t->w + k);
      w[k] = wk.x; w[k + 1] = wk.y;
    }
  }
  __syncthreads();
  if (x >= Wout) return;
  for (int r = 0; r < nrow; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < RESIZE_TAPS; ++k)
      if (k < count) acc += w[k] * (float)srow[r][start + k];
    dst[((long)pl * H + y0 + r) * Wout + x] = acc;
  }
}

template <class T>
void launch_resize(const T* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst,
                   hipStream_t st) {
  const ResizePlanRef p = resize_plan(Hin, Win, Hout, Wout, antialias);
  static_assert(sizeof(ResizeTable) == 8 + 4 * RESIZE_TAPS && RESIZE_TAPS % 2 == 0 && alignof(ResizeTable) >= 4 &&
                sizeof(ResizeTable) % 8 == 0, "the row kernel reads the weights as float2 pairs at offset 8 of 40-byte entries");
  if (Hin == Hout && (long)Win * 4 <= (long)Wout * 5 && planes <= 65535) {
    hipLaunchKernelGGL((resize_rows_kernel<T>), dim3((unsigned)cdiv(Wout, 256), (unsigned)cdiv(Hout, RZ_ROWS), (unsigned)planes), dim3(256),
                       0, st, src, p.tx, Hin, Win, Wout, dst);
    ATDN_HIP(hipGetLastError());
    return;
  }
  const long rows = (long)planes * Hout;
  // (grid.y is limited to 65,535 blocks: clips beyond that many rows go in slices of whole planes)
  const int planes_per = (int)std::max<long>(1, std::min<long>(planes, 65535 / Hout));
  for (int p0 = 0; p0 < planes; p0 += planes_per) {
    const int np = std::min(planes_per, planes - p0);
    const dim3 grid((unsigned)cdiv(Wout, 256), (unsigned)(np * Hout));
    const T* s0 = src + (long)p0 * Hin * Win;
    float* d0 = dst + (long)p0 * Hout * Wout;
    if (Hin == Hout) hipLaunchKernelGGL((resize_kernel<T, true>), grid, dim3(256), 0, st, s0, p.ty, p.tx, np, Hin, Win, Hout, Wout, d0);
    else hipLaunchKernelGGL((resize_kernel<T, false>), grid, dim3(256), 0, st, s0, p.ty, p.tx, np, Hin, Win, Hout, Wout, d0);
    ATDN_HIP(hipGetLastError());
  }
  (void)rows;
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
