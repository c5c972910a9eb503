// One CLVO training iteration on the device (train_odometry.py:21-49 per batch):
// ATDNVO in train mode over the T frames of B clips, CLVO_Loss (alpha = 1), back-propagation through time and
// through the convolutional encoder, AdamW. Gradients live in one flat buffer so that data-parallel training is a
// single all-reduce (RCCL) over it between `forward_backward` and `adamw_step`.
#pragma once
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "conv_dispatch.h"
#include "gma.h"  // DeviceBuf
#include "train_kernels.h"
#include "weights.h"

namespace atdn {

class ClvoTrainer {
 public:
  ClvoTrainer(int H, int W, int B, int T);
  ~ClvoTrainer();
  StateDict& state() { return sd_; }
  void finalize();

  // flows [B][T][2][H][W] (device, the reference's batch layout), targets [B][T][3] (device).
  // Fills the gradient buffer (overwritten, not accumulated), updates the BatchNorm running statistics, returns the
  // loss; pred_rot / pred_tr [B][T][3] (device, optional).
  float forward_backward(const float* flows, const float* true_rot, const float* true_tr, float* pred_rot, float* pred_tr,
                         hipStream_t st);
  // AdamW over every parameter that received a gradient (polar_norm is never used by forward()); t = 1-based step
  void adamw_step(float lr, float wd, float eps, int t, hipStream_t st);

  float* grad_buffer() { return grads_.p; }
  long grad_count() const { return n_params_; }
  // copy a named tensor to the host: kind 0 = parameter, 1 = gradient, 2 = BatchNorm running statistic
  long read(const std::string& key, int kind, float* host, long capacity, hipStream_t st);

  int H, W, B, T;

 private:
  struct Slot { long off = -1, n = 0; };
  struct ConvL {   // convolution + its packed forward / transposed weights
    Slot w, b;
    int cin = 16, cpix = 16, kh = 3, kw = 3, stride = 1, pad = 1;
    long fwd_off = 0, bwd_off = 0;  // into packed_
  };
  struct BnL { Slot gamma, beta; long rm = -1, rv = -1; long stat_off = 0; };  // stat_off: mean/rstd [G][16] x 2
  struct ConvBlock { ConvL conv; BnL bn; };
  struct ResBlock { ConvBlock a, b; ConvL skip; BnL out; };
  struct Lin { Slot w, b; int in = 0, out = 0; };

  Slot param(const std::string& key);
  long stat(const std::string& key);
  ConvL make_conv(const std::string& p, int stride, int pad);
  BnL make_bn(const std::string& p);
  Lin make_lin(const std::string& p, bool bias = true);

  void pack_weights(hipStream_t st);
  int conv_fwd(const ConvL& c, const float* x, int h, int w, float* z, hipStream_t st, bool with_stats = false);
  void conv_bwd_data(const ConvL& c, const float* dz, int h_in, int w_in, int ho, int wo, float* dx, int ldd, hipStream_t st,
                     bool accumulate = false);
  int bn_fwd(const BnL& bn, const float* z, long P, bool mish, const float* add, float* y, hipStream_t st, int rows_done = 0,
             bool stats_next = false);
  // dy -> dz (through BN and the activation); adds dgamma/dbeta; bias gradient of the producing conv into db (optional)
  void bn_bwd(const BnL& bn, const float* dy, const float* z, long P, bool mish, float* dz, float* db, hipStream_t st);

  float* P(const Slot& s) { return params_.p + s.off; }
  float* G(const Slot& s) { return grads_.p + s.off; }

  StateDict sd_;
  bool ready_ = false;
  bool conv16_ = !(getenv("ATDN_TRAIN_CONV16") && getenv("ATDN_TRAIN_CONV16")[0] == '0');  // 16-channel convs on the 16x16x4 MFMA kernel
  // BatchNorm statistics taken in the kernel that writes the layer's input (0: a reduction pass of their own, the A/B partner)
  bool fused_stats_ = !(getenv("ATDN_TRAIN_FUSED_STATS") && getenv("ATDN_TRAIN_FUSED_STATS")[0] == '0');
  std::map<std::string, Slot> pindex_;
  std::map<std::string, long> sindex_;
  long n_params_ = 0, n_stats_ = 0;
  std::vector<std::pair<long, long>> trained_;  // (offset, count) ranges AdamW walks
  DeviceBuf params_, grads_, m_, v_, stats_, packed_, bnstat_, part_, sums_, wscratch_, loss_;

  Slot dw_w_, dw_b_;
  ConvBlock stem_, last_;
  ResBlock res_[4];
  Lin fc_, lin_, rot_[3], tr_[3];
  Slot l1_wih_, l1_whh_, l1_bih_, l1_bhh_, l2_wih_, l2_whh_, l2_bih_, l2_bhh_;
  int hs_[7], ws_[7];  // map sizes: [0] input, [1] stem out, [2..5] res outs, [6] last conv out
  long packed_n_ = 0, bnstat_n_ = 0;

  // activations kept for the backward pass
  DeviceBuf flow_, x0_, z1_, y1_;
  struct ResAct { DeviceBuf za, ua, zb, s, zo, o; } ract_[4];
  DeviceBuf z6_, y6_, flat_, zf_, feat_;
  DeviceBuf pre1_, act1_, c1_, tc1_, h1_, zl_, x2_, pre2_, act2_, c2_, tc2_, h2_;
  DeviceBuf hz_[2][2], ha_[2][2], out_[2];  // heads: [rot|tr][layer]
  // gradient scratch
  DeviceBuf ga_, gb_, gc_, stuffed_, dsmall_[8];
};

}  // namespace atdn
