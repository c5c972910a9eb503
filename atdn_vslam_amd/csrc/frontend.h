// Frame front-end: resize (antialiased or plain bilinear, uint8 or fp32 source), replicate padding, and the
// host-uint8 -> device-fp32 ingest with its own copy stream (frontend.hip).
#pragma once
#include "common.h"

namespace atdn {

constexpr int RESIZE_TAPS = 8;
struct ResizeTable { int start; int count; float w[RESIZE_TAPS]; };
struct ResizePlanRef { const ResizeTable* ty; const ResizeTable* tx; };

// weight tables of one geometry on the current device (cached, thread-safe)
const ResizePlanRef resize_plan(int Hin, int Win, int Hout, int Wout, int antialias);

// src [planes][Hin][Win] (float or unsigned char) -> dst [planes][Hout][Wout] fp32
template <class T>
void launch_resize(const T* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst, hipStream_t st);

void launch_pad_replicate(const float* src, int planes, int H, int W, int l, int r, int t, int b, float* dst, hipStream_t st);

class FrameIngest {
 public:
  FrameIngest(int Hin, int Win, int Hout, int Wout, int max_frames, int antialias);
  ~FrameIngest();
  FrameIngest(const FrameIngest&) = delete;
  FrameIngest& operator=(const FrameIngest&) = delete;
  // host_frames: n x [3][Hin][Win] uint8 (pinned memory makes the copy asynchronous) -> dst n x [3][Hout][Wout] fp32
  void ingest(const unsigned char* host_frames, int n, float* dst, hipStream_t st);
  const int Hin, Win, Hout, Wout, max_frames, antialias;

 private:
  unsigned char* stage_[2] = {nullptr, nullptr};
  hipEvent_t copied_[2] = {nullptr, nullptr}, consumed_[2] = {nullptr, nullptr};
  bool used_[2] = {false, false};
  int next_ = 0;
  int dev_ = 0;
  hipStream_t copy_stream_ = nullptr;
};

}  // namespace atdn
