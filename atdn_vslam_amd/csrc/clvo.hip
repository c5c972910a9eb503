// CLVO pose head on MI355X. Reference: atdn_vslam/odometry/network.py:63-73,122-146;
// layers/conv.py:36-37,83-90; layers/linear.py:35-42; utils/normalizations.py:8-10.
#include "clvo.h"
#include "train_kernels.h"
#include <cstdlib>

namespace atdn {

extern template TileChoice conv_dispatch<MODE_TAP, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);

ClvoNet::ClvoNet(int H_, int W_, int max_batch) : H(H_), W(W_), maxB(max_batch) {
  ATDN_CHECK(max_batch >= 1 && max_batch <= 256, "max_batch out of range");
  // the head flattens a 16x4x13 map into Linear(832) (odometry/network.py:70-72): only some frame sizes fit
  int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
  for (int i = 0; i < 4; ++i) { h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1); }
  h = conv_out(h, 3, 3, 0); w = conv_out(w, 3, 3, 0);
  if (h * w * 16 != 832) {
    char b[160];
    snprintf(b, sizeof b, "ATDNVO needs a flow size that reduces to 16x4x13 (got 16x%dx%d from %dx%d)", h, w, H, W);
    throw Error(b);
  }
  ATDN_HIP(hipGetDevice(&dev_));   // (after the argument checks: those must report without a device)
}

ClvoNet::~ClvoNet() {
  DeviceGuard dg(dev_);           // the handle's device, not whichever is current
  (void)hipDeviceSynchronize();   // a scan graph of this handle may still be running on the caller's stream
  for (auto& kv : scan_graphs_) (void)hipGraphExecDestroy(kv.second);
  if (cap_stream_) (void)hipStreamDestroy(cap_stream_);
  if (scan_abort_host_) (void)hipHostFree(scan_abort_host_);
  for (DeviceBuf* b : {&in4_, &bufA_, &bufB_, &bufS_, &flat_, &pre_, &hseq_, &x2seq_, &hseq2_, &cstate_, &scan_xch_, &scan_state0_}) b->release();
  arena_.release();
}

ClvoNet::ConvBN ClvoNet::pack_convbn(const std::string& p) {
  ConvBN c;
  const int cin = (int)sd_.get(p + ".conv.weight").shape[1];
  c.conv = pack_conv(arena_, sd_, {p + ".conv"}, MODE_ROW, cin <= 4 ? 4 : 16);
  ChannelAffine a = bn_affine(sd_, p + ".bn");
  std::vector<float> sc(a.scale.begin(), a.scale.end()), sh(a.shift.begin(), a.shift.end());
  c.sc_off = pack_vector(arena_, sc);
  c.sh_off = pack_vector(arena_, sh);
  c.raw_w_off = pack_vector(arena_, sd_.get(p + ".conv.weight").data);
  c.raw_b_off = pack_vector(arena_, sd_.get(p + ".conv.bias").data);
  return c;
}

ClvoNet::Lin ClvoNet::pack_linear(const std::string& wkey, const std::string& bkey, const std::vector<int>* perm) {
  Lin l;
  const HostTensor& w = sd_.get(wkey);
  const int N = (int)w.shape[0], K = (int)w.shape[1];
  l.w_off = arena_.alloc((long)N * K);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) arena_.at(l.w_off)[(long)n * K + k] = w.data[(long)n * K + (perm ? (*perm)[k] : k)];
  if (!bkey.empty()) l.b_off = pack_vector(arena_, sd_.get(bkey).data);
  return l;
}

void ClvoNet::finalize() {
  ATDN_CHECK(!ready_, "finalize called twice");
  dw_w_off_ = pack_vector(arena_, sd_.get("encoder_CNN.0.weight").data);
  dw_b_off_ = pack_vector(arena_, sd_.get("encoder_CNN.0.bias").data);
  stem_ = pack_convbn("encoder_CNN.1");
  for (int i = 0; i < 4; ++i) {
    const std::string p = "encoder_CNN." + std::to_string(i + 2);
    res_[i].a = pack_convbn(p + ".conv.0");
    res_[i].b = pack_convbn(p + ".conv.1");
    res_[i].skip = pack_conv(arena_, sd_, {p + ".skip_layer"}, MODE_ROW, 16);
    res_[i].skip_w_off = pack_vector(arena_, sd_.get(p + ".skip_layer.weight").data);
    res_[i].skip_b_off = pack_vector(arena_, sd_.get(p + ".skip_layer.bias").data);
    ChannelAffine a = bn_affine(sd_, p + ".out_block.1");
    std::vector<float> sc(a.scale.begin(), a.scale.end()), sh(a.shift.begin(), a.shift.end());
    res_[i].sc_off = pack_vector(arena_, sc);
    res_[i].sh_off = pack_vector(arena_, sh);
  }
  last_ = pack_convbn("encoder_CNN.6");
  // Flatten is (C,H,W)-ordered in the reference; our activations are (H,W,C): permute the FC columns once
  std::vector<int> perm(832);
  for (int p = 0; p < 52; ++p)
    for (int c = 0; c < 16; ++c) perm[p * 16 + c] = c * 52 + p;
  fc_ = pack_linear("encoder_CNN.8.linear.weight", "encoder_CNN.8.linear.bias", &perm);
  lstm1_ih_ = pack_linear("lstm1.weight_ih", "lstm1.bias_ih");
  lstm1_hh_ = pack_linear("lstm1.weight_hh", "lstm1.bias_hh");
  lstm_lin_ = pack_linear("lstm_linear.linear.weight", "lstm_linear.linear.bias");
  lstm2_ih_ = pack_linear("lstm2.weight_ih", "lstm2.bias_ih");
  lstm2_hh_ = pack_linear("lstm2.weight_hh", "lstm2.bias_hh");
  const char* heads[2] = {"rotation_regressor", "translation_regressor"};
  for (int hd = 0; hd < 2; ++hd) {
    Lin* L = hd ? tr_ : rot_;
    const std::string p = heads[hd];
    L[0] = pack_linear(p + ".0.linear.weight", p + ".0.linear.bias");
    L[1] = pack_linear(p + ".1.linear.weight", p + ".1.linear.bias");
    L[2] = pack_linear(p + ".2.weight", "");
  }
  arena_.upload();
  auto fix = [&](ConvBN& c) {
    resolve(arena_, c.conv); c.sc = arena_.dev(c.sc_off); c.sh = arena_.dev(c.sh_off);
    c.raw_w = arena_.dev(c.raw_w_off); c.raw_b = arena_.dev(c.raw_b_off);
  };
  fix(stem_); fix(last_);
  for (auto& r : res_) {
    fix(r.a); fix(r.b); resolve(arena_, r.skip); r.sc = arena_.dev(r.sc_off); r.sh = arena_.dev(r.sh_off);
    r.skip_w = arena_.dev(r.skip_w_off); r.skip_b = arena_.dev(r.skip_b_off);
  }
  scan_graph_ = !(getenv("ATDN_NO_GRAPH") && getenv("ATDN_NO_GRAPH")[0] == '1');
  scan_persistent_ = !(getenv("ATDN_SCAN_PERSISTENT") && getenv("ATDN_SCAN_PERSISTENT")[0] == '0') && lstm_scan_fits_device();
  scan_xch_.alloc((lstm_scan_exchange_bytes() + 3) / 4);
  scan_state0_.alloc(4 * 512);
  ATDN_HIP(hipHostMalloc(reinterpret_cast<void**>(&scan_abort_host_), 64, hipHostMallocDefault));
  *scan_abort_host_ = 0u;
  ATDN_HIP(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
  for (Lin* l : {&fc_, &lstm1_ih_, &lstm1_hh_, &lstm_lin_, &lstm2_ih_, &lstm2_hh_, &rot_[0], &rot_[1], &rot_[2],
                 &tr_[0], &tr_[1], &tr_[2]}) {
    l->w = arena_.dev(l->w_off);
    l->b = l->b_off >= 0 ? arena_.dev(l->b_off) : nullptr;
  }
  const int h1 = conv_out(H, 7, 2, 3), w1 = conv_out(W, 7, 2, 3);
  in4_.alloc((long)maxB * H * W * 4);
  bufA_.alloc((long)maxB * h1 * w1 * 16);
  bufB_.alloc((long)maxB * h1 * w1 * 16);
  bufS_.alloc((long)maxB * h1 * w1 * 16 / 4 + 64);
  flat_.alloc((long)maxB * 832);
  ready_ = true;
}

void ClvoNet::encode(const float* flow, int B, float* feat, hipStream_t st) {
  ATDN_CHECK(ready_, "weights not finalized");
  ATDN_CHECK(B >= 1 && B <= maxB, "batch exceeds max_batch of this handle");
  launch_prep_flow(flow, B, H, W, arena_.dev(dw_w_off_), arena_.dev(dw_b_off_), in4_.p, st);
  int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
  float* x = bufA_.p; float* t = bufB_.p;
  {
    // every layer after the depthwise 1x1 is 16 channels wide: the 16x16x4 fp32 MFMA kernels (train_kernels.hip) hold
    // the whole weight tensor in operand registers and their N tile is exactly 16 columns — a 32x32x2 tile is a quarter
    // full on these layers. Eval-mode BatchNorm/Mish (and the residual tail) are fused into the store.
    Conv16Tail ts; ts.sc = stem_.sc; ts.sh = stem_.sh;
    launch_stem16(in4_.p, B, H, W, stem_.raw_w, stem_.raw_b, x, st, &ts);
    for (int i = 0; i < 4; ++i) {
      const Res& r = res_[i];
      const int oh = conv_out(h, 3, 2, 1), ow = conv_out(w, 3, 2, 1);
      Conv16Tail ta; ta.sc = r.a.sc; ta.sh = r.a.sh;
      launch_conv16_eval(x, B, h, w, r.a.raw_w, r.a.raw_b, 3, 1, 1, ta, t, st);
      launch_conv16(x, B, h, w, r.skip_w, false, r.skip_b, 1, 2, 0, bufS_.p, st);
      Conv16Tail tb; tb.sc = r.b.sc; tb.sh = r.b.sh; tb.skip = bufS_.p; tb.sc2 = r.sc; tb.sh2 = r.sh;
      launch_conv16_eval(t, B, h, w, r.b.raw_w, r.b.raw_b, 3, 2, 1, tb, x, st);   // x is dead after the skip conv
      h = oh; w = ow;
    }
    Conv16Tail tl; tl.sc = last_.sc; tl.sh = last_.sh;
    launch_conv16_eval(x, B, h, w, last_.raw_w, last_.raw_b, 3, 3, 0, tl, flat_.p, st);
    launch_linear(fc_.w, flat_.p, 832, 832, nullptr, nullptr, 0, 0, fc_.b, nullptr, 1, feat, 512, 512, B, st);
  }
}

void ClvoNet::ensure_scan(long rows, int Bs) {
  bool grown = false;
  auto grow = [&](DeviceBuf& b, long n) { if (b.n < n) { b.release(); b.alloc(n); grown = true; } };
  grow(pre_, rows * 2048);
  grow(hseq_, (rows + Bs) * 512);
  grow(x2seq_, rows * 512);
  grow(hseq2_, (rows + Bs) * 512);
  grow(cstate_, 2L * Bs * 512);
  if (grown) {   // captured graphs hold the old addresses
    (void)hipDeviceSynchronize();
    for (auto& kv : scan_graphs_) (void)hipGraphExecDestroy(kv.second);
    scan_graphs_.clear();
  }
}

// the T + 2 launches of the three-stage pipeline (lstm1 step s, lstm_linear step s - 1, lstm2 step s - 2)
void ClvoNet::launch_scan_steps(int T, int Bs, hipStream_t st) {
  const long sb = (long)Bs * 512;
  float* c1 = cstate_.p; float* c2 = cstate_.p + sb;
  for (int sidx = 0; sidx < T + 2; ++sidx) {
    LstmPipeArgs a{};
    a.Hd = 512; a.B = Bs;
    a.do1 = sidx < T; a.do_lin = sidx >= 1 && sidx <= T; a.do2 = sidx >= 2;
    const int t1 = sidx < T ? sidx : 0, tl = a.do_lin ? sidx - 1 : 0, t2 = a.do2 ? sidx - 2 : 0;
    a.pre1 = pre_.p + (long)t1 * Bs * 2048; a.Whh1 = lstm1_hh_.w; a.bhh1 = lstm1_hh_.b;
    a.h1_in = hseq_.p + t1 * sb; a.c1 = c1; a.h1_out = hseq_.p + (t1 + 1) * sb;
    a.Wlin = lstm_lin_.w; a.blin = lstm_lin_.b; a.lin_in = hseq_.p + (tl + 1) * sb; a.lin_out = x2seq_.p + tl * sb;
    a.Wih2 = lstm2_ih_.w; a.bih2 = lstm2_ih_.b; a.Whh2 = lstm2_hh_.w; a.bhh2 = lstm2_hh_.b;
    a.x2_in = x2seq_.p + t2 * sb; a.h2_in = hseq2_.p + t2 * sb; a.c2 = c2; a.h2_out = hseq2_.p + (t2 + 1) * sb;
    launch_lstm_pipe(a, st);
  }
}

// The recurrence of odometry/network.py:137-140 restructured so that only what is truly sequential stays in the
// per-step loop. lstm1 does not depend on lstm2, so: (1) input projections of ALL steps in one MFMA GEMM,
// (2) scan 1: one fused kernel per step (W_hh·h + gates + cell), (3) lstm_linear + lstm2 input projection batched
// over all steps, (4) scan 2, (5) both regressors batched. 2 small launches per step instead of 6.
void ClvoNet::step(const float* feat, int T, int Bs, float* state, float* rot, float* tr, hipStream_t st) {
  ATDN_CHECK(ready_, "weights not finalized");
  ATDN_CHECK(Bs >= 1 && Bs <= maxB && T >= 1, "bad sequence shape");
  const long rows = (long)T * Bs;
  ensure_scan(rows, Bs);
  float* h1 = state; float* c1 = state + (long)Bs * 512; float* h2 = state + 2L * Bs * 512; float* c2 = state + 3L * Bs * 512;
  const long sb = (long)Bs * 512;
  auto gemm = [&](const float* x, const Lin& L, float* y, int N) {  // y[rows][N] = x[rows][512]·W^T + b on the MFMA engine
    ConvShape s;
    s.src0 = x; s.ld0 = 512; s.sb0 = 0; s.C0 = 512; s.H = 1; s.W = (int)rows;
    s.w = L.w; s.ldw = 512; s.N = N; s.nimg = 1;
    conv_dispatch<MODE_TAP>(s, EpiBias<ACT_NONE>{L.b, y, 0, N, 1.f}, st);
  };
  gemm(feat, lstm1_ih_, pre_.p, 2048);
  const MlpHead R{rot_[0].w, rot_[0].b, rot_[1].w, rot_[1].b, rot_[2].w};
  const MlpHead Tt{tr_[0].w, tr_[0].b, tr_[1].w, tr_[1].b, tr_[2].w};
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &capturing);
  if (scan_persistent_ && Bs == 1 && T >= kPersistentMinSteps && capturing == hipStreamCaptureStatusNone) {
    // ONE launch for the whole sequence (lstm_scan.hip): reads and writes the caller's state in place, h2 of every step into
    // hseq2_ rows 1..T (where the per-step pipeline leaves them), then both regressors batched over the sequence.
    // The launch is VERIFIED before its results are used: the stream is synchronised (a sequence scan is the end of a
    // sequence: its caller fetches the poses next) and the abort word read; a launch that gave up on a bounded spin —
    // workgroups that never became resident together — is repeated on the per-step kernel from the saved state, and the handle
    // stays there. Nobody ever sees the NaN poses the kernel leaves behind on that path.
    ATDN_HIP(hipMemcpyAsync(scan_state0_.p, state, 4 * 512 * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_lstm_scan(pre_.p, lstm1_hh_.w, lstm1_hh_.b, lstm_lin_.w, lstm_lin_.b, lstm2_ih_.w, lstm2_ih_.b, lstm2_hh_.w,
                     lstm2_hh_.b, state, hseq2_.p + sb, scan_xch_.p, T, st);
    ATDN_HIP(hipMemcpyAsync(scan_abort_host_, reinterpret_cast<const char*>(scan_xch_.p) + (lstm_scan_exchange_bytes() - 64),
                            sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    ATDN_HIP(hipStreamSynchronize(st));
    if (*scan_abort_host_ == 0u) {
      launch_mlp_heads(hseq2_.p + sb, (int)rows, R, Tt, rot, tr, st);
      return;
    }
    fprintf(stderr, "atdn: the persistent LSTM scan gave up on a bounded spin (its workgroups were not resident together); "
                    "repeating the sequence on the per-step kernel and staying there\n");
    scan_persistent_ = false;
    *scan_abort_host_ = 0u;
    ATDN_HIP(hipMemcpyAsync(state, scan_state0_.p, 4 * 512 * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  {
    // lstm1 (step s), lstm_linear (step s - 1) and lstm2 with its input projection (step s - 2) share ONE launch per
    // step: T + 2 dependent launches instead of 2T + 2, replayed as one hipGraph per (T, Bs)
    ATDN_HIP(hipMemcpyAsync(hseq_.p, h1, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    ATDN_HIP(hipMemcpyAsync(hseq2_.p, h2, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    ATDN_HIP(hipMemcpyAsync(cstate_.p, c1, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    ATDN_HIP(hipMemcpyAsync(cstate_.p + sb, c2, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    // Graph policy: a (T, Bs) is graphed only when it is seen for the SECOND time and T is moderate. A one-off
    // sequence (run_sequence scans all pairs of a KITTI sequence in one call, T ~ 4500) would pay capture + instantiation
    // of a multi-thousand-node graph for a single replay; the step is bound by its dependent launches' L2 round trips
    // either way, so eager launches lose nothing there. A caller that repeats one length (bench.py) warms it TWICE:
    // first sight runs eagerly, second sight captures, the timed call replays.
    const auto key = std::make_pair(T, Bs);
    const bool seen = scan_seen_.count(key) != 0;
    if (!seen && scan_seen_.size() < 4096) scan_seen_.insert(key);
    if (scan_graph_ && T >= 4 && T <= kMaxGraphedSteps && (seen || scan_graphs_.count(key))) {
      if (!scan_graphs_.count(key)) {
        hipGraph_t graph = nullptr;
        ATDN_HIP(hipStreamBeginCapture(cap_stream_, hipStreamCaptureModeThreadLocal));
        try {
          launch_scan_steps(T, Bs, cap_stream_);
        } catch (...) {
          (void)hipStreamEndCapture(cap_stream_, &graph);
          if (graph) (void)hipGraphDestroy(graph);
          throw;
        }
        ATDN_HIP(hipStreamEndCapture(cap_stream_, &graph));
        hipGraphExec_t exec = nullptr;
        ATDN_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (scan_graphs_.size() >= 16) {   // bounded cache: sequences of many different lengths
          (void)hipDeviceSynchronize();
          for (auto& kv : scan_graphs_) (void)hipGraphExecDestroy(kv.second);
          scan_graphs_.clear();
        }
        scan_graphs_[key] = exec;
      }
      ATDN_HIP(hipGraphLaunch(scan_graphs_[key], st));
    } else {
      launch_scan_steps(T, Bs, st);
    }
    ATDN_HIP(hipMemcpyAsync(c1, cstate_.p, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    ATDN_HIP(hipMemcpyAsync(c2, cstate_.p + sb, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    ATDN_HIP(hipMemcpyAsync(h1, hseq_.p + (long)T * sb, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    ATDN_HIP(hipMemcpyAsync(h2, hseq2_.p + (long)T * sb, sb * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_mlp_heads(hseq2_.p + sb, (int)rows, R, Tt, rot, tr, st);
  }
}

}  // namespace atdn
