// CLVO training iteration. Reference: train_odometry.py:21-49 (loop body), odometry/network.py:122-146 (forward, train
// mode), layers/conv.py (Conv = BN(Mish(conv)), ResidualConv), layers/linear.py, odometry/loss.py:25-58,108-118.
// Convolutions (forward and data gradients) run on the exact-fp32 MFMA implicit-GEMM engine; a strided convolution's
// data gradient is the stride-1 convolution of the zero-stuffed output gradient with the transposed, flipped kernel.
#include "clvo_train.h"

#include <algorithm>
#include <cstring>
#include <utility>

#include "kernels.h"

namespace atdn {

extern template TileChoice conv_dispatch<MODE_ROW, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);

namespace {
// [B][T][...] <-> [T][B][...] with `inner` contiguous floats per (b, t) item
__global__ void swap_bt_kernel(const float* __restrict__ src, int B, int T, long inner, int src_is_bt, float* __restrict__ dst,
                               long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long item = i / inner, e = i - item * inner;
    long s;
    if (src_is_bt) { const long t = item / B, b = item - t * B; s = (b * T + t) * inner + e; }   // dst is [T][B]
    else { const long b = item / T, t = item - b * T; s = (t * B + b) * inner + e; }             // dst is [B][T]
    dst[i] = src[s];
  }
}
void launch_swap_bt(const float* src, int B, int T, long inner, bool src_is_bt, float* dst, hipStream_t st) {
  const long total = (long)B * T * inner;
  hipLaunchKernelGGL(swap_bt_kernel, dim3((unsigned)std::min<long>(cdivl(total, 256), 65535)), dim3(256), 0, st, src, B, T,
                     inner, src_is_bt ? 1 : 0, dst, total);
  ATDN_HIP(hipGetLastError());
}
bool is_stat(const std::string& k) {
  auto ends = [&](const char* s) { const size_t n = strlen(s); return k.size() >= n && k.compare(k.size() - n, n, s) == 0; };
  return ends("running_mean") || ends("running_var");
}
bool is_counter(const std::string& k) { return k.size() >= 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0; }
}  // namespace

ClvoTrainer::ClvoTrainer(int H_, int W_, int B_, int T_) : H(H_), W(W_), B(B_), T(T_) {
  ATDN_CHECK(B >= 2 && B <= 64 && T >= 1 && T <= 32, "batch / sequence length out of range (BatchNorm needs B >= 2)");
  hs_[0] = H; ws_[0] = W;
  hs_[1] = conv_out(H, 7, 2, 3); ws_[1] = conv_out(W, 7, 2, 3);
  for (int i = 2; i <= 5; ++i) { hs_[i] = conv_out(hs_[i - 1], 3, 2, 1); ws_[i] = conv_out(ws_[i - 1], 3, 2, 1); }
  hs_[6] = conv_out(hs_[5], 3, 3, 0); ws_[6] = conv_out(ws_[5], 3, 3, 0);
  ATDN_CHECK(hs_[6] * ws_[6] * 16 == 832, "ATDNVO needs a flow size that reduces to a 16x4x13 map");
}

ClvoTrainer::~ClvoTrainer() {
  for (DeviceBuf* b : {&params_, &grads_, &m_, &v_, &stats_, &packed_, &bnstat_, &part_, &sums_, &wscratch_, &loss_, &flow_, &x0_,
                       &z1_, &y1_, &z6_, &y6_, &flat_, &zf_, &feat_, &pre1_, &act1_, &c1_, &tc1_, &h1_, &zl_, &x2_, &pre2_, &act2_,
                       &c2_, &tc2_, &h2_, &out_[0], &out_[1], &ga_, &gb_, &gc_, &stuffed_})
    b->release();
  for (auto& r : ract_) for (DeviceBuf* b : {&r.za, &r.ua, &r.zb, &r.s, &r.zo, &r.o}) b->release();
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { hz_[i][j].release(); ha_[i][j].release(); }
  for (auto& b : dsmall_) b.release();
}

ClvoTrainer::Slot ClvoTrainer::param(const std::string& key) {
  auto it = pindex_.find(key);
  if (it == pindex_.end()) throw Error("missing parameter: " + key);
  return it->second;
}
long ClvoTrainer::stat(const std::string& key) {
  auto it = sindex_.find(key);
  if (it == sindex_.end()) throw Error("missing running statistic: " + key);
  return it->second;
}
ClvoTrainer::ConvL ClvoTrainer::make_conv(const std::string& p, int stride, int pad) {
  ConvL c;
  c.w = param(p + ".weight");
  c.b = param(p + ".bias");
  const HostTensor& w = sd_.get(p + ".weight");
  ATDN_CHECK((int)w.shape[0] == 16, "encoder convolutions have 16 output channels");
  c.cin = (int)w.shape[1]; c.kh = (int)w.shape[2]; c.kw = (int)w.shape[3];
  c.cpix = c.cin <= 4 ? 4 : 16;
  c.stride = stride; c.pad = pad;
  c.fwd_off = packed_n_;
  packed_n_ += (long)16 * c.kh * round_up(c.kw * c.cpix, 32);
  c.bwd_off = packed_n_;
  packed_n_ += (long)c.cin * c.kh * round_up(c.kw * 16, 32);
  return c;
}
ClvoTrainer::BnL ClvoTrainer::make_bn(const std::string& p) {
  BnL b;
  b.gamma = param(p + ".weight");
  b.beta = param(p + ".bias");
  b.rm = stat(p + ".running_mean");
  b.rv = stat(p + ".running_var");
  b.stat_off = bnstat_n_;
  bnstat_n_ += 2L * T * 16;
  return b;
}
ClvoTrainer::Lin ClvoTrainer::make_lin(const std::string& p, bool bias) {
  Lin l;
  l.w = param(p + ".weight");
  if (bias) l.b = param(p + ".bias");
  const HostTensor& w = sd_.get(p + ".weight");
  l.out = (int)w.shape[0]; l.in = (int)w.shape[1];
  return l;
}

void ClvoTrainer::finalize() {
  ATDN_CHECK(!ready_, "finalize called twice");
  // ---- registry: every float tensor that is not a running statistic is a parameter, in key order
  std::vector<std::string> keys = sd_.keys();
  std::vector<float> hp, hs;
  for (const std::string& k : keys) {
    if (is_counter(k)) continue;
    const HostTensor& t = sd_.get(k);
    if (is_stat(k)) {
      sindex_[k] = (long)hs.size();
      hs.insert(hs.end(), t.data.begin(), t.data.end());
      while (hs.size() % 4) hs.push_back(0.f);
    } else {
      Slot s; s.off = (long)hp.size(); s.n = t.numel();
      pindex_[k] = s;
      hp.insert(hp.end(), t.data.begin(), t.data.end());
      while (hp.size() % 4) hp.push_back(0.f);
      if (k.rfind("polar_norm.", 0) != 0) trained_.push_back({s.off, s.n});   // forward() never touches polar_norm
    }
  }
  n_params_ = (long)hp.size(); n_stats_ = (long)hs.size();
  params_.alloc(n_params_); grads_.alloc(n_params_); m_.alloc(n_params_); v_.alloc(n_params_); stats_.alloc(n_stats_);
  ATDN_HIP(hipMemcpy(params_.p, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
  ATDN_HIP(hipMemcpy(stats_.p, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
  ATDN_HIP(hipMemset(grads_.p, 0, n_params_ * sizeof(float)));
  ATDN_HIP(hipMemset(m_.p, 0, n_params_ * sizeof(float)));
  ATDN_HIP(hipMemset(v_.p, 0, n_params_ * sizeof(float)));

  dw_w_ = param("encoder_CNN.0.weight"); dw_b_ = param("encoder_CNN.0.bias");
  stem_.conv = make_conv("encoder_CNN.1.conv", 2, 3); stem_.bn = make_bn("encoder_CNN.1.bn");
  for (int i = 0; i < 4; ++i) {
    const std::string p = "encoder_CNN." + std::to_string(i + 2);
    res_[i].a.conv = make_conv(p + ".conv.0.conv", 1, 1); res_[i].a.bn = make_bn(p + ".conv.0.bn");
    res_[i].b.conv = make_conv(p + ".conv.1.conv", 2, 1); res_[i].b.bn = make_bn(p + ".conv.1.bn");
    res_[i].skip = make_conv(p + ".skip_layer", 2, 0);
    res_[i].out = make_bn(p + ".out_block.1");
  }
  last_.conv = make_conv("encoder_CNN.6.conv", 3, 0); last_.bn = make_bn("encoder_CNN.6.bn");
  fc_ = make_lin("encoder_CNN.8.linear");
  lin_ = make_lin("lstm_linear.linear");
  l1_wih_ = param("lstm1.weight_ih"); l1_whh_ = param("lstm1.weight_hh"); l1_bih_ = param("lstm1.bias_ih"); l1_bhh_ = param("lstm1.bias_hh");
  l2_wih_ = param("lstm2.weight_ih"); l2_whh_ = param("lstm2.weight_hh"); l2_bih_ = param("lstm2.bias_ih"); l2_bhh_ = param("lstm2.bias_hh");
  const char* heads[2] = {"rotation_regressor", "translation_regressor"};
  for (int hd = 0; hd < 2; ++hd) {
    Lin* L = hd ? tr_ : rot_;
    L[0] = make_lin(std::string(heads[hd]) + ".0.linear");
    L[1] = make_lin(std::string(heads[hd]) + ".1.linear");
    L[2] = make_lin(std::string(heads[hd]) + ".2", false);
  }

  // ---- buffers
  const long nimg = (long)T * B, TB = nimg;
  auto px = [&](int l) { return (long)hs_[l] * ws_[l]; };
  packed_.alloc(packed_n_);
  bnstat_.alloc(bnstat_n_);
  const long Pmax = (long)B * px(1);
  // partial rows of the BatchNorm reductions: per group, one per reduction block, per persistent block of a producing convolution
  // (<= 1024: train_kernels.hip) or per block of bn_apply
  part_.alloc((long)T * std::max<long>(std::max<long>(bn_partial_blocks(Pmax), bn_apply_partial_rows(Pmax)), 1024) * 2 * 16 + 64);
  sums_.alloc((long)T * 2 * 16);
  wscratch_.alloc(std::max(wgrad_scratch_floats(0, 0, 16, 3, 3), wgrad_scratch_floats(0, 0, 3, 7, 7)));
  loss_.alloc(4);
  flow_.alloc(nimg * 2 * px(0));
  x0_.alloc(nimg * px(0) * 4);
  z1_.alloc(nimg * px(1) * 16); y1_.alloc(nimg * px(1) * 16);
  for (int k = 0; k < 4; ++k) {
    const long in = nimg * px(k + 1) * 16, out = nimg * px(k + 2) * 16;
    ract_[k].za.alloc(in); ract_[k].ua.alloc(in);
    ract_[k].zb.alloc(out); ract_[k].s.alloc(out); ract_[k].zo.alloc(out); ract_[k].o.alloc(out);
  }
  z6_.alloc(nimg * 832); y6_.alloc(nimg * 832); flat_.alloc(nimg * 832);
  zf_.alloc(TB * 512); feat_.alloc(TB * 512);
  pre1_.alloc(TB * 2048); act1_.alloc(TB * 2048); c1_.alloc((TB + B) * 512); tc1_.alloc(TB * 512); h1_.alloc((TB + B) * 512);
  zl_.alloc(TB * 512); x2_.alloc(TB * 512);
  pre2_.alloc(TB * 2048); act2_.alloc(TB * 2048); c2_.alloc((TB + B) * 512); tc2_.alloc(TB * 512); h2_.alloc((TB + B) * 512);
  for (int hd = 0; hd < 2; ++hd) {
    hz_[hd][0].alloc(TB * 128); ha_[hd][0].alloc(TB * 128);
    hz_[hd][1].alloc(TB * 64); ha_[hd][1].alloc(TB * 64);
    out_[hd].alloc(TB * 4);
  }
  const long gmax = nimg * px(1) * 16;
  ga_.alloc(gmax); gb_.alloc(gmax); gc_.alloc(gmax);
  stuffed_.alloc(nimg * px(1) * 16);  // largest zero-stuffed gradient: a stride-2 block's output at its input size
  // small gradient scratch: 0 d_rot, 1 d_tr, 2 dh2 [TB][512], 3 dpre [TB][2048], 4 dh1 / dx [TB][512], 5 rec + dc ping-pong,
  // 6 head scratch, 7 dflat [TB][832]
  dsmall_[0].alloc(TB * 4); dsmall_[1].alloc(TB * 4); dsmall_[2].alloc(TB * 512); dsmall_[3].alloc(TB * 2048);
  dsmall_[4].alloc(TB * 512); dsmall_[5].alloc(3L * B * 512); dsmall_[6].alloc(TB * 256);
  dsmall_[7].alloc(std::max<long>(TB * 832, 16 * 3 * 49));
  ready_ = true;
}

long ClvoTrainer::read(const std::string& key, int kind, float* host, long capacity, hipStream_t st) {
  ATDN_CHECK(ready_, "not finalized");
  const float* src; long n;
  if (kind == 2) {
    const HostTensor& t = sd_.get(key);
    src = stats_.p + stat(key); n = t.numel();
  } else {
    const Slot s = param(key);
    src = (kind == 0 ? params_.p : grads_.p) + s.off; n = s.n;
  }
  ATDN_CHECK(host && capacity >= n, "read: destination too small");
  ATDN_HIP(hipMemcpyAsync(host, src, n * sizeof(float), hipMemcpyDeviceToHost, st));
  ATDN_HIP(hipStreamSynchronize(st));
  return n;
}

// ---------------------------------------------------------------------------------------------------- building blocks
void ClvoTrainer::pack_weights(hipStream_t st) {
  auto pack = [&](const ConvL& c) {
    launch_pack_row(P(c.w), 16, c.cin, c.cpix, c.kh, c.kw, false, packed_.p + c.fwd_off, st);
    launch_pack_row(P(c.w), 16, c.cin, 16, c.kh, c.kw, true, packed_.p + c.bwd_off, st);
  };
  pack(stem_.conv); pack(last_.conv);
  for (auto& r : res_) { pack(r.a.conv); pack(r.b.conv); pack(r.skip); }
}

// Returns the partial rows per group of the BatchNorm statistics of Mish(z) the kernel left in part_ (with_stats, 16-channel
// kernels only), or 0: the caller's bn_fwd then reduces z itself.
int ClvoTrainer::conv_fwd(const ConvL& c, const float* x, int h, int w, float* z, hipStream_t st, bool with_stats) {
  Conv16Stats cs;
  cs.part = with_stats && fused_stats_ ? part_.p : nullptr; cs.group_imgs = B; cs.capacity = part_.n;
  if (c.cin == 16 && conv16_) {   // 16 -> 16 channels: the 16x16x4 fp32 MFMA kernel, weights straight from the parameters
    launch_conv16(x, T * B, h, w, P(c.w), false, P(c.b), c.kh, c.stride, c.pad, z, st, false, &cs);
    return cs.rows;
  }
  if (c.cin == 2 && c.cpix == 4 && c.kh == 7 && c.kw == 7 && c.stride == 2 && c.pad == 3 && conv16_) {
    launch_stem16(x, T * B, h, w, P(c.w), P(c.b), z, st, nullptr, &cs);
    return cs.rows;
  }
  ConvShape s;
  s.src0 = x; s.ld0 = c.cpix; s.sb0 = (long)h * w * c.cpix; s.C0 = c.cpix; s.H = h; s.W = w;
  s.KH = c.kh; s.KW = c.kw; s.stride = c.stride; s.padH = c.pad; s.padW = c.pad;
  s.w = packed_.p + c.fwd_off; s.ldw = c.kh * round_up(c.kw * c.cpix, 32); s.N = 16; s.nimg = T * B;
  const int oh = conv_out(h, c.kh, c.stride, c.pad), ow = conv_out(w, c.kw, c.stride, c.pad);
  conv_dispatch<MODE_ROW>(s, EpiBias<ACT_NONE>{P(c.b), z, (long)oh * ow * 16, 16, 1.f}, st);
  return 0;
}

void ClvoTrainer::conv_bwd_data(const ConvL& c, const float* dz, int h_in, int w_in, int ho, int wo, float* dx, int ldd,
                                hipStream_t st, bool accumulate) {
  if (c.cin == 16 && ldd == 16 && conv16_ && c.stride == 2) {   // no zero-stuffed map: parity classes of the output
    launch_tconv16_s2(dz, T * B, ho, wo, P(c.w), c.kh, c.pad, h_in, w_in, accumulate, dx, st);
    return;
  }
  const int Hs = h_in - c.kh + 1 + 2 * c.pad, Ws = w_in - c.kw + 1 + 2 * c.pad;
  const float* src = dz;
  if (c.stride > 1) {
    ATDN_CHECK((long)T * B * Hs * Ws * 16 <= stuffed_.n, "zero-stuffing buffer too small");
    launch_zero_stuff(dz, T * B, ho, wo, c.stride, Hs, Ws, stuffed_.p, st);
    src = stuffed_.p;
  } else {
    ATDN_CHECK(Hs == ho && Ws == wo, "stride-1 data gradient expects same-size maps");
  }
  if (c.cin == 16 && ldd == 16 && conv16_) {
    ATDN_CHECK(Hs + c.kh - 1 - 2 * c.pad == h_in && Ws + c.kw - 1 - 2 * c.pad == w_in, "data-gradient geometry");
    launch_conv16(src, T * B, Hs, Ws, P(c.w), true, nullptr, c.kh, 1, c.kh - 1 - c.pad, dx, st, accumulate);
    return;
  }
  ConvShape s;
  s.src0 = src; s.ld0 = 16; s.sb0 = (long)Hs * Ws * 16; s.C0 = 16; s.H = Hs; s.W = Ws;
  s.KH = c.kh; s.KW = c.kw; s.stride = 1; s.padH = c.kh - 1 - c.pad; s.padW = c.kw - 1 - c.pad;
  s.w = packed_.p + c.bwd_off; s.ldw = c.kh * round_up(c.kw * 16, 32); s.N = c.cin; s.nimg = T * B;
  ATDN_CHECK(conv_out(Hs, c.kh, 1, s.padH) == h_in && conv_out(Ws, c.kw, 1, s.padW) == w_in, "data-gradient geometry");
  ATDN_CHECK(!accumulate, "conv_bwd_data: accumulation needs the 16-channel kernels");
  conv_dispatch<MODE_ROW>(s, EpiBias<ACT_NONE>{nullptr, dx, (long)h_in * w_in * ldd, ldd, 1.f}, st);
}

// rows_done > 0: part_ already holds that many partial rows per group of this layer's statistics (left there by the kernel that
// produced z). stats_next: this pass also leaves the statistics of Mish(y) in part_ for the BatchNorm that follows y directly;
// returns their rows per group (part_ is free again by then: the finalize that read it is earlier in the stream).
int ClvoTrainer::bn_fwd(const BnL& bn, const float* z, long Pg, bool mish, const float* add, float* y, hipStream_t st, int rows_done,
                        bool stats_next) {
  float* mean = bnstat_.p + bn.stat_off;
  float* rstd = mean + (long)T * 16;
  if (rows_done > 0) {
    ATDN_CHECK(mish, "fused statistics are those of Mish(z)");
    launch_bn_finalize_rows(part_.p, T, rows_done, Pg, stats_.p + bn.rm, stats_.p + bn.rv, mean, rstd, sums_.p, st);
  } else {
    launch_bn_stats(z, T, Pg, mish, part_.p, st);
    launch_bn_finalize(part_.p, T, Pg, stats_.p + bn.rm, stats_.p + bn.rv, mean, rstd, sums_.p, st);   // (sums_: idle in the forward)
  }
  const bool next = stats_next && fused_stats_;
  launch_bn_apply(z, T, Pg, mish, mean, rstd, P(bn.gamma), P(bn.beta), add, y, st, next ? part_.p : nullptr);
  return next ? bn_apply_partial_rows(Pg) : 0;
}

void ClvoTrainer::bn_bwd(const BnL& bn, const float* dy, const float* z, long Pg, bool mish, float* dz, float* db, hipStream_t st) {
  const float* mean = bnstat_.p + bn.stat_off;
  const float* rstd = mean + (long)T * 16;
  launch_bn_bwd_stats(dy, z, T, Pg, mish, mean, rstd, part_.p, st);
  launch_bn_bwd_finalize(part_.p, T, Pg, sums_.p, G(bn.gamma), G(bn.beta), st);
  launch_bn_bwd_apply(dy, z, T, Pg, mish, mean, rstd, P(bn.gamma), sums_.p, dz, part_.p, st);
  if (db) launch_sum_partials16(part_.p, (long)T * bn_partial_blocks(Pg), db, st);
}

// ---------------------------------------------------------------------------------------------------- one iteration
float ClvoTrainer::forward_backward(const float* flows, const float* true_rot, const float* true_tr, float* pred_rot,
                                    float* pred_tr, hipStream_t st) {
  ATDN_CHECK(ready_, "weights not finalized");
  const int nimg = T * B, TB = nimg;
  auto px = [&](int l) { return (long)hs_[l] * ws_[l]; };
  auto Pg = [&](int l) { return (long)B * px(l); };
  ATDN_HIP(hipMemsetAsync(grads_.p, 0, n_params_ * sizeof(float), st));
  pack_weights(st);

  // ================= forward (train mode)
  launch_swap_bt(flows, B, T, 2 * px(0), true, flow_.p, st);   // [B][T] clip layout -> step-major [T][B]
  launch_prep_flow(flow_.p, nimg, H, W, P(dw_w_), P(dw_b_), x0_.p, st);
  // (round 5: every BatchNorm's statistics come out of the kernel that writes its input — the convolution, or for out_block the
  // pass that forms zo — instead of a reduction pass of their own; part_ carries them to the finalize right behind)
  int rows = conv_fwd(stem_.conv, x0_.p, H, W, z1_.p, st, true);
  bn_fwd(stem_.bn, z1_.p, Pg(1), true, nullptr, y1_.p, st, rows);
  const float* x = y1_.p;
  for (int k = 0; k < 4; ++k) {
    ResBlock& r = res_[k]; ResAct& A = ract_[k];
    const int h = hs_[k + 1], w = ws_[k + 1];
    rows = conv_fwd(r.a.conv, x, h, w, A.za.p, st, true);
    bn_fwd(r.a.bn, A.za.p, Pg(k + 1), true, nullptr, A.ua.p, st, rows);
    rows = conv_fwd(r.b.conv, A.ua.p, h, w, A.zb.p, st, true);
    conv_fwd(r.skip, x, h, w, A.s.p, st, false);
    rows = bn_fwd(r.b.bn, A.zb.p, Pg(k + 2), true, A.s.p, A.zo.p, st, rows, true);    // zo = BN_b(mish(zb)) + skip
    bn_fwd(r.out, A.zo.p, Pg(k + 2), true, nullptr, A.o.p, st, rows);
    x = A.o.p;
  }
  rows = conv_fwd(last_.conv, x, hs_[5], ws_[5], z6_.p, st, true);
  bn_fwd(last_.bn, z6_.p, Pg(6), true, nullptr, y6_.p, st, rows);
  launch_nhwc_to_chw(y6_.p, nimg, (int)px(6), flat_.p, st);         // nn.Flatten order (C, H, W)
  launch_gemm(false, true, TB, 512, 832, flat_.p, 832, P(fc_.w), 832, zf_.p, 512, 0.f, P(fc_.b), st);
  launch_mish_fwd(zf_.p, feat_.p, (long)TB * 512, st);

  const long sb = (long)B * 512;
  auto lstm_forward = [&](const float* xin, const Slot& wih, const Slot& whh, const Slot& bih, const Slot& bhh, DeviceBuf& pre,
                          DeviceBuf& act, DeviceBuf& c, DeviceBuf& tc, DeviceBuf& hbuf) {
    ATDN_HIP(hipMemsetAsync(c.p, 0, sb * sizeof(float), st));
    ATDN_HIP(hipMemsetAsync(hbuf.p, 0, sb * sizeof(float), st));
    launch_gemm(false, true, TB, 2048, 512, xin, 512, P(wih), 512, pre.p, 2048, 0.f, P(bih), st);
    for (int t = 0; t < T; ++t) {
      float* p = pre.p + (long)t * B * 2048;
      launch_gemm(false, true, B, 2048, 512, hbuf.p + t * sb, 512, P(whh), 512, p, 2048, 1.f, P(bhh), st);
      launch_lstm_fwd(p, c.p + t * sb, B, act.p + (long)t * B * 2048, c.p + (t + 1) * sb, tc.p + t * sb, hbuf.p + (t + 1) * sb, st);
    }
  };
  lstm_forward(feat_.p, l1_wih_, l1_whh_, l1_bih_, l1_bhh_, pre1_, act1_, c1_, tc1_, h1_);
  launch_gemm(false, true, TB, 512, 512, h1_.p + sb, 512, P(lin_.w), 512, zl_.p, 512, 0.f, P(lin_.b), st);
  launch_mish_fwd(zl_.p, x2_.p, (long)TB * 512, st);
  lstm_forward(x2_.p, l2_wih_, l2_whh_, l2_bih_, l2_bhh_, pre2_, act2_, c2_, tc2_, h2_);
  const float* h2out = h2_.p + sb;
  for (int hd = 0; hd < 2; ++hd) {
    Lin* L = hd ? tr_ : rot_;
    launch_gemm(false, true, TB, 128, 512, h2out, 512, P(L[0].w), 512, hz_[hd][0].p, 128, 0.f, P(L[0].b), st);
    launch_mish_fwd(hz_[hd][0].p, ha_[hd][0].p, (long)TB * 128, st);
    launch_gemm(false, true, TB, 64, 128, ha_[hd][0].p, 128, P(L[1].w), 128, hz_[hd][1].p, 64, 0.f, P(L[1].b), st);
    launch_mish_fwd(hz_[hd][1].p, ha_[hd][1].p, (long)TB * 64, st);
    launch_gemm(false, true, TB, 3, 64, ha_[hd][1].p, 64, P(L[2].w), 64, out_[hd].p, 3, 0.f, nullptr, st);
  }
  if (pred_rot) launch_swap_bt(out_[0].p, B, T, 3, false, pred_rot, st);
  if (pred_tr) launch_swap_bt(out_[1].p, B, T, 3, false, pred_tr, st);

  // ================= loss and backward
  float* d_out[2] = {dsmall_[0].p, dsmall_[1].p};
  launch_clvo_loss(out_[0].p, out_[1].p, true_rot, true_tr, B, T, loss_.p, d_out[0], d_out[1], st);
  float* dh2 = dsmall_[2].p;
  float* hs1 = dsmall_[6].p;               // [TB][128]
  float* hs2 = dsmall_[6].p + (long)TB * 128;  // [TB][64]
  for (int hd = 0; hd < 2; ++hd) {
    Lin* L = hd ? tr_ : rot_;
    launch_gemm(true, false, 3, 64, TB, d_out[hd], 3, ha_[hd][1].p, 64, G(L[2].w), 64, 1.f, nullptr, st);      // dW2 += dout^T a1
    launch_gemm(false, false, TB, 64, 3, d_out[hd], 3, P(L[2].w), 64, hs2, 64, 0.f, nullptr, st);              // da1 = dout W2
    launch_mish_bwd(hs2, hz_[hd][1].p, hs2, (long)TB * 64, st);
    launch_gemm(true, false, 64, 128, TB, hs2, 64, ha_[hd][0].p, 128, G(L[1].w), 128, 1.f, nullptr, st);
    launch_colsum(hs2, TB, 64, 64, G(L[1].b), st);
    launch_gemm(false, false, TB, 128, 64, hs2, 64, P(L[1].w), 128, hs1, 128, 0.f, nullptr, st);
    launch_mish_bwd(hs1, hz_[hd][0].p, hs1, (long)TB * 128, st);
    launch_gemm(true, false, 128, 512, TB, hs1, 128, h2out, 512, G(L[0].w), 512, 1.f, nullptr, st);
    launch_colsum(hs1, TB, 128, 128, G(L[0].b), st);
    launch_gemm(false, false, TB, 512, 128, hs1, 128, P(L[0].w), 512, dh2, 512, hd ? 1.f : 0.f, nullptr, st);
  }
  // dh_ext [T][B][512] (modified in place) -> dx [T][B][512]; parameter gradients accumulated
  auto lstm_backward = [&](float* dh_ext, const float* xin, const Slot& wih, const Slot& whh, const Slot& bih, const Slot& bhh,
                           DeviceBuf& act, DeviceBuf& c, DeviceBuf& tc, DeviceBuf& hbuf, float* dx) {
    float* dpre = dsmall_[3].p;
    float* rec = dsmall_[5].p;
    float* dc[2] = {dsmall_[5].p + sb, dsmall_[5].p + 2 * sb};
    for (int t = T - 1; t >= 0; --t) {
      float* dh = dh_ext + t * sb;
      if (t < T - 1) launch_add_inplace(dh, rec, sb, st);
      launch_lstm_bwd(dh, t < T - 1 ? dc[(t + 1) & 1] : nullptr, act.p + (long)t * B * 2048, c.p + t * sb, tc.p + t * sb, B,
                      dpre + (long)t * B * 2048, dc[t & 1], st);
      if (t > 0) launch_gemm(false, false, B, 512, 2048, dpre + (long)t * B * 2048, 2048, P(whh), 512, rec, 512, 0.f, nullptr, st);
    }
    launch_gemm(true, false, 2048, 512, TB, dpre, 2048, xin, 512, G(wih), 512, 1.f, nullptr, st);
    launch_gemm(true, false, 2048, 512, TB, dpre, 2048, hbuf.p, 512, G(whh), 512, 1.f, nullptr, st);   // h[0..T-1]
    launch_colsum(dpre, TB, 2048, 2048, G(bih), st);
    launch_colsum(dpre, TB, 2048, 2048, G(bhh), st);
    launch_gemm(false, false, TB, 512, 2048, dpre, 2048, P(wih), 512, dx, 512, 0.f, nullptr, st);
  };
  float* dx2 = dsmall_[4].p;
  lstm_backward(dh2, x2_.p, l2_wih_, l2_whh_, l2_bih_, l2_bhh_, act2_, c2_, tc2_, h2_, dx2);
  launch_mish_bwd(dx2, zl_.p, dx2, (long)TB * 512, st);                                                  // dzl
  launch_gemm(true, false, 512, 512, TB, dx2, 512, h1_.p + sb, 512, G(lin_.w), 512, 1.f, nullptr, st);
  launch_colsum(dx2, TB, 512, 512, G(lin_.b), st);
  float* dh1 = dsmall_[2].p;
  launch_gemm(false, false, TB, 512, 512, dx2, 512, P(lin_.w), 512, dh1, 512, 0.f, nullptr, st);
  float* dfeat = dsmall_[4].p;
  lstm_backward(dh1, feat_.p, l1_wih_, l1_whh_, l1_bih_, l1_bhh_, act1_, c1_, tc1_, h1_, dfeat);
  launch_mish_bwd(dfeat, zf_.p, dfeat, (long)TB * 512, st);                                              // dzf
  launch_gemm(true, false, 512, 832, TB, dfeat, 512, flat_.p, 832, G(fc_.w), 832, 1.f, nullptr, st);
  launch_colsum(dfeat, TB, 512, 512, G(fc_.b), st);
  launch_gemm(false, false, TB, 832, 512, dfeat, 512, P(fc_.w), 832, dsmall_[7].p, 832, 0.f, nullptr, st);  // dflat (C,H,W)

  // ---- convolutional encoder, last layer to first. Three map-sized buffers rotate: cur (incoming gradient), t1, t2
  float* cur = ga_.p; float* t1 = gb_.p; float* t2 = gc_.p;
  launch_chw_to_nhwc(dsmall_[7].p, nimg, (int)px(6), cur, st);
  bn_bwd(last_.bn, cur, z6_.p, Pg(6), true, cur, G(last_.conv.b), st);
  launch_conv_wgrad(ract_[3].o.p, 16, 16, nimg, hs_[5], ws_[5], cur, hs_[6], ws_[6], 3, 3, 3, 0, wscratch_.p, G(last_.conv.w), st);
  conv_bwd_data(last_.conv, cur, hs_[5], ws_[5], hs_[6], ws_[6], t1, 16, st);
  std::swap(cur, t1);
  for (int k = 3; k >= 0; --k) {
    ResBlock& r = res_[k]; ResAct& A = ract_[k];
    const int h = hs_[k + 1], w = ws_[k + 1], oh = hs_[k + 2], ow = ws_[k + 2];
    const float* xin = k == 0 ? y1_.p : ract_[k - 1].o.p;
    bn_bwd(r.out, cur, A.zo.p, Pg(k + 2), true, cur, G(r.skip.b), st);                      // cur = dzo; db_skip = sum dzo
    launch_conv_wgrad(xin, 16, 16, nimg, h, w, cur, oh, ow, 1, 1, 2, 0, wscratch_.p, G(r.skip.w), st);
    conv_bwd_data(r.skip, cur, h, w, oh, ow, t2, 16, st);                                   // t2 = dx through the skip conv
    bn_bwd(r.b.bn, cur, A.zb.p, Pg(k + 2), true, cur, G(r.b.conv.b), st);                   // cur = dzb
    launch_conv_wgrad(A.ua.p, 16, 16, nimg, h, w, cur, oh, ow, 3, 3, 2, 1, wscratch_.p, G(r.b.conv.w), st);
    conv_bwd_data(r.b.conv, cur, h, w, oh, ow, t1, 16, st);                                 // t1 = dua
    bn_bwd(r.a.bn, t1, A.za.p, Pg(k + 1), true, t1, G(r.a.conv.b), st);                     // t1 = dza
    launch_conv_wgrad(xin, 16, 16, nimg, h, w, t1, h, w, 3, 3, 1, 1, wscratch_.p, G(r.a.conv.w), st);
    if (conv16_) {
      conv_bwd_data(r.a.conv, t1, h, w, h, w, t2, 16, st, true);                            // t2 += dx through conv a
      std::swap(cur, t2);
    } else {
      conv_bwd_data(r.a.conv, t1, h, w, h, w, cur, 16, st);                                 // cur = dx through conv a
      launch_add_inplace(cur, t2, (long)nimg * h * w * 16, st);
    }
  }
  bn_bwd(stem_.bn, cur, z1_.p, Pg(1), true, cur, G(stem_.conv.b), st);                      // cur = dz1
  // stem weights and the depthwise 1x1 in front of it: one weight-gradient pass on (xn0, xn1, 1), then a combine
  launch_prep_flow_aux(flow_.p, nimg, H, W, t1, st);                                        // t1 = auxiliary input, NHWC4
  float* A = dsmall_[7].p;                                                                  // [16][3][49]
  ATDN_HIP(hipMemsetAsync(A, 0, 16 * 3 * 49 * sizeof(float), st));
  launch_conv_wgrad(t1, 4, 3, nimg, H, W, cur, hs_[1], ws_[1], 7, 7, 2, 3, wscratch_.p, A, st);
  launch_stem_combine(A, P(stem_.conv.w), P(dw_w_), P(dw_b_), 49, G(stem_.conv.w), G(dw_w_), G(dw_b_), st);

  float loss = 0.f;
  ATDN_HIP(hipMemcpyAsync(&loss, loss_.p, sizeof(float), hipMemcpyDeviceToHost, st));
  ATDN_HIP(hipStreamSynchronize(st));
  return loss;
}

void ClvoTrainer::adamw_step(float lr, float wd, float eps, int t, hipStream_t st) {
  ATDN_CHECK(ready_ && t >= 1, "adamw_step: bad state");
  for (auto& r : trained_)
    launch_adamw(params_.p + r.first, grads_.p + r.first, m_.p + r.first, v_.p + r.first, r.second, lr, wd, eps, 0.9f, 0.999f, t, st);
}

}  // namespace atdn
