// MappingVAE encoder (atdn_vslam/localization/network.py:29-45,57-70): the embedding `mu` that NeuralSLAM's
// relocalisation compares between a query frame and the stored keyframes. Decoder and training stay outside.
#pragma once
#include "conv_dispatch.h"
#include "kernels.h"
#include "weights.h"
#include "gma.h"  // DeviceBuf

namespace atdn {

class VaeEncoder {
 public:
  VaeEncoder(int H, int W, int max_batch);
  ~VaeEncoder();
  StateDict& state() { return sd_; }
  void finalize();
  // images NCHW [B,3,H,W] with values 0..255 -> mu NHWC [B][h*w][128] (h, w = out_h(), out_w())
  void encode(const float* images, int B, float* mu, hipStream_t st);
  int out_h() const { return oh_; }
  int out_w() const { return ow_; }

  int H, W, maxB;

 private:
  struct ConvBN { PackedConv conv; long sc_off = -1, sh_off = -1; const float* sc = nullptr; const float* sh = nullptr; };
  struct Res { ConvBN a, b; PackedConv skip; long sc_off = -1, sh_off = -1; const float* sc = nullptr; const float* sh = nullptr; };
  ConvBN pack_convbn(const std::string& p);

  StateDict sd_;
  WeightArena arena_;
  bool ready_ = false;
  ConvBN stem_;
  Res res_[6];
  PackedConv mean_;
  DeviceBuf in4_, bufA_, bufB_, bufS_;
  int oh_ = 0, ow_ = 0;
};

}  // namespace atdn
