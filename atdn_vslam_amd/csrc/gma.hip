// GMA optical-flow forward on MI355X: encoders, all-pairs correlation pyramid, GMA attention, 12x
// (lookup -> motion encoder -> attention aggregate -> separable ConvGRU -> flow head), mask head and
// convex upsampling.  Reference: whl:GMA/core/network.py:72-129 and the blocks it calls.
#include "gma.h"
#include "small_convs.h"
#include "conv_sf.h"
#include "epilogues_sf.h"

namespace atdn { const float* zero_line(); }

namespace atdn {

extern template TileChoice conv_dispatch<MODE_TAP, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiBias<ACT_RELU>>(const ConvShape&, EpiBias<ACT_RELU>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_ROW, EpiBias<ACT_RELU>>(const ConvShape&, EpiBias<ACT_RELU>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiBiasStats>(const ConvShape&, EpiBiasStats, hipStream_t);
extern template TileChoice conv_dispatch<MODE_ROW, EpiBiasStats>(const ConvShape&, EpiBiasStats, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiBiasReluAddRelu>(const ConvShape&, EpiBiasReluAddRelu, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiContextSplit>(const ConvShape&, EpiContextSplit, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiScale>(const ConvShape&, EpiScale, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiQK>(const ConvShape&, EpiQK, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiStoreT>(const ConvShape&, EpiStoreT, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiAggregate>(const ConvShape&, EpiAggregate, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiGruZR>(const ConvShape&, EpiGruZR, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiGruQ>(const ConvShape&, EpiGruQ, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiFlowDelta>(const ConvShape&, EpiFlowDelta, hipStream_t);

extern template TileChoice conv_dispatch<MODE_ROW, SfBias<ACT_RELU>>(const ConvShape&, SfBias<ACT_RELU>, hipStream_t);
#define ATDN_EXTERN_SF(EPI) extern template TileChoice conv_sf_dispatch<EPI>(const ConvShape&, float, EPI, hipStream_t);
ATDN_EXTERN_SF(SfBias<ACT_NONE>) ATDN_EXTERN_SF(SfBias<ACT_RELU>) ATDN_EXTERN_SF(EpiBias<ACT_NONE>)
ATDN_EXTERN_SF(EpiBiasStats) ATDN_EXTERN_SF(SfBiasReluAddRelu) ATDN_EXTERN_SF(SfContextSplit)
ATDN_EXTERN_SF(SfQK) ATDN_EXTERN_SF(SfVT)
ATDN_EXTERN_SF(SfGruZR) ATDN_EXTERN_SF(SfGruQ)
extern template void conv_sf_dispatch_pair<SfBias<ACT_RELU>>(const ConvShape&, float, SfBias<ACT_RELU>, const ConvShape&, float,
                                                             SfBias<ACT_RELU>, hipStream_t);

namespace {

// precision mode 2: the sf convolutions of this thread issue only the hi x hi MFMA while the guard lives (conv_sf.h)
struct FastGuard {
  bool prev;
  explicit FastGuard(bool on) : prev(sf_fast_mode()) { sf_fast_mode() = on; }
  ~FastGuard() { sf_fast_mode() = prev; }
};

constexpr int XLD = 384;       // GRU input x = [inp | motion(126) flow(2) | motion_global]  (update.py:130)
constexpr int CORR_LD = 352;   // 4*81 lookup channels padded to a multiple of 32

ConvShape conv_shape(const PackedConv& L, const float* src, int ld, long sb, int nimg, int H, int W, int stride,
                     int padH, int padW) {
  ConvShape s;
  s.src0 = src; s.ld0 = ld; s.sb0 = sb; s.C0 = L.C;
  s.H = H; s.W = W; s.KH = L.KH; s.KW = L.KW; s.stride = stride; s.padH = padH; s.padW = padW;
  s.w = L.w; s.wfrag16 = L.wf16; s.ldw = L.ldw; s.N = L.N; s.nimg = nimg;
  return s;
}

EncoderWeights pack_encoder(WeightArena& A, const StateDict& sd, const std::string& p, bool batchnorm, bool sf) {
  EncoderWeights E;
  auto fold = [&](const std::string& norm) { return bn_affine(sd, norm); };
  auto pack_conv = [&](WeightArena& A_, const StateDict& sd_, const std::vector<std::string>& names, int mode, int cpix,
                       const ChannelAffine* f = nullptr) {
    return (sf && mode == MODE_TAP) ? pack_conv_sf(A_, sd_, names, f) : atdn::pack_conv(A_, sd_, names, mode, cpix, f);
  };
  if (batchnorm) { auto a = fold(p + "norm1"); E.stem = pack_conv(A, sd, {p + "conv1"}, MODE_ROW, 4, &a); }
  else E.stem = pack_conv(A, sd, {p + "conv1"}, MODE_ROW, 4);
  if (sf) pack_stem_sf(A, E.stem);
  int bi = 0;
  for (int li = 1; li <= 3; ++li)
    for (int k = 0; k < 2; ++k, ++bi) {
      const std::string q = p + "layer" + std::to_string(li) + "." + std::to_string(k) + ".";
      auto& B = E.blk[bi];
      B.has_ds = (li > 1 && k == 0);
      if (batchnorm) {
        auto a1 = fold(q + "norm1"), a2 = fold(q + "norm2");
        B.c1 = pack_conv(A, sd, {q + "conv1"}, MODE_TAP, 0, &a1);
        B.c2 = pack_conv(A, sd, {q + "conv2"}, MODE_TAP, 0, &a2);
        if (B.has_ds) { auto a3 = fold(q + "norm3"); B.ds = pack_conv(A, sd, {q + "downsample.0"}, MODE_TAP, 0, &a3); }
      } else {
        B.c1 = pack_conv(A, sd, {q + "conv1"}, MODE_TAP, 0);
        B.c2 = pack_conv(A, sd, {q + "conv2"}, MODE_TAP, 0);
        if (B.has_ds) B.ds = pack_conv(A, sd, {q + "downsample.0"}, MODE_TAP, 0);
      }
    }
  E.head = pack_conv(A, sd, {p + "conv2"}, MODE_TAP, 0);
  return E;
}

void resolve_encoder(const WeightArena& A, EncoderWeights& E) {
  resolve(A, E.stem);
  for (auto& b : E.blk) { resolve(A, b.c1); resolve(A, b.c2); if (b.has_ds) resolve(A, b.ds); }
  resolve(A, E.head);
}

}  // namespace

struct GmaNet::Timer {
  std::vector<std::pair<int, hipEvent_t>> marks;  // (stage that ENDS at this event)
  hipEvent_t start = nullptr;
};

void GmaNet::mark(int stage, hipStream_t st) {
  if (!timer_) return;
  hipEvent_t e;
  ATDN_HIP(hipEventCreate(&e));
  ATDN_HIP(hipEventRecord(e, st));
  timer_->marks.emplace_back(stage, e);
}

void GmaNet::profile(int B, int iters, int reps, float* ms, hipStream_t st, int mode) {
  ATDN_CHECK(ready_ && B >= 1 && B <= maxB && reps >= 1, "bad profile request");
  ATDN_CHECK(mode == 0 || precision >= 1, "sequence modes are built for the split-f16 pipeline");
  for (int i = 0; i < ST_COUNT; ++i) ms[i] = 0.f;
  seq_ = mode;   // 0 pair, 1 sequence, 2 continued sequence (fmap_ slot 0 is read as it stands: timing only)
  last_frame_ = 0;   // fmap_ is overwritten: a continued sequence call must not read it
  for (int r = 0; r < reps; ++r) {
    Timer t;
    timer_ = &t;
    if (precision >= 1) launch_init_coords_sf(nullptr, B, H8, W8, coords1_.p, flow4_.p, x_.p, XLD, 254, st);
    else launch_init_coords(nullptr, B, H8, W8, coords1_.p, flow4_.p, x_.p + 254, XLD, st);
    ATDN_HIP(hipEventCreate(&t.start));
    ATDN_HIP(hipEventRecord(t.start, st));
    try { if (precision >= 1) { FastGuard fg(precision == 2); run_body_sf(B, iters, st); } else run_body(B, iters, st); } catch (...) { timer_ = nullptr; throw; }
    timer_ = nullptr;
    ATDN_HIP(hipStreamSynchronize(st));
    hipEvent_t prev = t.start;
    for (auto& m : t.marks) {
      float d = 0.f;
      ATDN_HIP(hipEventElapsedTime(&d, prev, m.second));
      ms[m.first] += d;
      prev = m.second;
    }
    for (auto& m : t.marks) (void)hipEventDestroy(m.second);
    (void)hipEventDestroy(t.start);
  }
}

GmaNet::GmaNet(int H_, int W_, int max_batch, int precision_) : H(H_), W(W_), maxB(max_batch), precision(precision_) {
  ATDN_CHECK(precision >= 0 && precision <= 2, "precision must be 0 (fp32 MFMA), 1 (split-f16 MFMA) or 2 (plain f16 MFMA)");
  ATDN_CHECK(H % 8 == 0 && W % 8 == 0 && H >= 64 && W >= 64, "frame size must be a multiple of 8 (use the padder)");
  ATDN_CHECK(max_batch >= 1 && max_batch <= 64, "max_batch out of range");
  H8 = H / 8; W8 = W / 8; N = H8 * W8; ldN = round_up(N, 32);
  norm_on_load_ = precision == 1;   // (the f16 fast mode keeps the separate normalisation pass: its conv kernels are FAST builds)
  classic_ = precision == 0;        // exact-fp32 mode: row-major pyramid + separate lookup, logits + softmax pass + GEMM
  const char* ng = getenv("ATDN_NO_GRAPH");
  use_graph_ = !(ng && ng[0] == '1');
  (void)hipGetDevice(&dev_);   // (no throw: argument errors must be reportable without a device; finalize() needs one anyway)
}

void GmaNet::fork(hipStream_t from, hipStream_t to) {
  ATDN_CHECK(par_next_ < par_events_.size(), "out of branch events");
  hipEvent_t e = par_events_[par_next_++];
  ATDN_HIP(hipEventRecord(e, from));
  ATDN_HIP(hipStreamWaitEvent(to, e, 0));
}

void GmaNet::set_low_latency(bool on) {
  ATDN_CHECK(!ready_, "set_low_latency: call it before finalize() (the workspace and the captured graphs depend on it)");
  low_latency_ = on;
}

GmaNet::~GmaNet() {
  DeviceGuard dg(dev_);   // the handle's device, not whichever is current
  // a graph replay or kernel of this handle may still be running on the caller's stream
  (void)hipDeviceSynchronize();
  for (auto& kv : graphs_) (void)hipGraphExecDestroy(kv.second);
  if (cap_stream_) (void)hipStreamDestroy(cap_stream_);
  if (par_stream_) (void)hipStreamDestroy(par_stream_);
  for (auto& e : par_events_) (void)hipEventDestroy(e);
  DeviceBuf* all[] = {&img4_, &enc_[0], &enc_[1], &enc_[2], &enc_[3], &scratch_, &pcnt_, &fin_, &fmap_, &psum_, &pm2_, &mean_[0], &mean_[1], &mean_[2], &rstd_[0],
                      &rstd_[1], &rstd_[2], &pyr_[0], &pyr_[1], &pyr_[2], &pyr_[3], &h_[0], &h_[1], &x_, &qk_, &attn_, &vT_,
                      &corrfeat_, &cor1_, &corflo_, &flo1_, &z_, &rh_, &fh_, &mask_, &coords1_, &flow4_, &pre_zr_[0],
                      &pre_zr_[1], &pre_q_[0], &pre_q_[1], &rowmax_, &rinv_,
                      &fbrick_[0], &fbrick_[1], &fbrick_[2], &fbrick_[3], &fplain_[0], &fplain_[1], &fplain_[2], &coords_used_, &fhG_,
                      &attn_part_, &enc2_[0], &enc2_[1], &enc2_[2], &enc2_[3]};
  for (auto* b : all) b->release();
  arena_.release();
}

void GmaNet::finalize() {
  ATDN_CHECK(!ready_, "finalize called twice");
  const std::string u = "update_block.";
  const bool sf = precision >= 1;
  auto tap = [&](const std::vector<std::string>& names, bool has_bias = true) {
    return sf ? pack_conv_sf(arena_, sd_, names, nullptr, has_bias)
              : pack_conv(arena_, sd_, names, MODE_TAP, 0, nullptr, has_bias);
  };
  fnet_ = pack_encoder(arena_, sd_, "fnet.", false, sf);
  cnet_ = pack_encoder(arena_, sd_, "cnet.", true, sf);
  convc1_ = tap({u + "encoder.convc1"});
  if (sf) pack_fragment_major16(arena_, convc1_);   // the fused lookup kernel loads operand-order weights (16x16x32)
  convc2_ = tap({u + "encoder.convc2"});
  convf1_ = pack_conv(arena_, sd_, {u + "encoder.convf1"}, MODE_ROW, 4);
  {  // the same weights in split-f16 fragment order for flow_conv7_sf_kernel (small_convs.hip): K = 8 rows x (8 taps x 2
     // channels), the eighth row and tap zero; [wave][channel block][step][hi, lo][lane (n, g)] x 8 f16, lane (n, g) = channel
     // 32 wave + 16 block + n, filter row 2 step + (g >> 1), taps 4 (g & 1) .. + 3; scaled by 2^p so that max |w| is in [1, 2)
    const HostTensor& w = sd_.get(u + "encoder.convf1.weight");
    ATDN_CHECK(w.shape[0] == 128 && w.shape[1] == 2 && w.shape[2] == 7 && w.shape[3] == 7, "convf1 is 7x7, 2 -> 128");
    float mx = 0.f;
    for (float v : w.data) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f) (void)std::frexp(mx, &e);
    const int p = 1 - e;
    convf1_wscale_ = std::ldexp(1.0f, -p);
    convf1_sf_off_ = arena_.alloc(4L * 2 * 4 * 2 * 64 * 4);
    _Float16* f = reinterpret_cast<_Float16*>(arena_.at(convf1_sf_off_));
    for (int wv = 0; wv < 4; ++wv)
      for (int cb = 0; cb < 2; ++cb)
        for (int s = 0; s < 4; ++s)
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
              const int n = 32 * wv + 16 * cb + (lane & 15), g = lane >> 4;
              const int ky = 2 * s + (g >> 1), kx = 4 * (g & 1) + (i >> 1), c = i & 1;
              const float v = (ky < 7 && kx < 7) ? std::ldexp(w.data[(((long)n * 2 + c) * 7 + ky) * 7 + kx], p) : 0.f;
              const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
              const long frag = (((long)wv * 2 + cb) * 4 + s) * 2;
              f[(frag * 64 + lane) * 8 + i] = hi;
              f[((frag + 1) * 64 + lane) * 8 + i] = lo;
            }
  }
  convf2_ = tap({u + "encoder.convf2"});
  convm_ = tap({u + "encoder.conv"});
  to_v_ = tap({u + "aggregator.to_v"}, false);
  to_qk_ = tap({"att.to_qk"}, false);
  for (int p = 0; p < 2; ++p) {
    const std::string t = std::to_string(p + 1);
    if (sf) {
      // hx = [h(0:128) | inp(128:256) | motion(256:384) | motion_global(384:512)] (update.py:50,130): the inp slice
      // does not change over the iterations, so its contribution is computed once per pair (run_body_sf)
      const std::vector<std::pair<int, int>> rest = {{0, 128}, {256, 512}}, ctx = {{128, 256}};
      gru_zr_[p] = pack_conv_sf_channels(arena_, sd_, {u + "gru.convz" + t, u + "gru.convr" + t}, rest, true);
      gru_q_[p] = pack_conv_sf_channels(arena_, sd_, {u + "gru.convq" + t}, rest, true);
      gru_zr_ctx_[p] = pack_conv_sf_channels(arena_, sd_, {u + "gru.convz" + t, u + "gru.convr" + t}, ctx, false);
      gru_q_ctx_[p] = pack_conv_sf_channels(arena_, sd_, {u + "gru.convq" + t}, ctx, false);
    } else {
      gru_zr_[p] = tap({u + "gru.convz" + t, u + "gru.convr" + t});
      gru_q_[p] = tap({u + "gru.convq" + t});
    }
  }
  fh1_ = tap({u + "flow_head.conv1"});
  fh2_ = tap({u + "flow_head.conv2"});
  if (sf) {   // conv2 once more, as fp32 rows [tap * 2 + output][256] (the fused flow head multiplies in fp32)
    const HostTensor& w = sd_.get(u + "flow_head.conv2.weight");
    ATDN_CHECK(w.shape.size() == 4 && w.shape[0] == 2 && w.shape[1] == 256 && w.shape[2] == 3 && w.shape[3] == 3, "flow head conv2 shape");
    fh2_w32_off_ = arena_.alloc(18 * 256);
    for (int o = 0; o < 2; ++o)
      for (int c = 0; c < 256; ++c)
        for (int t = 0; t < 9; ++t) arena_.at(fh2_w32_off_)[(t * 2 + o) * 256 + c] = w.data[((long)o * 256 + c) * 9 + t];
    float mx = 0.f;
    for (float v : w.data) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f) (void)std::frexp(mx, &e);
    fh2_mul_ = std::ldexp(1.0f, 1 - e);
  }
  mask0_ = tap({u + "mask.0"});
  mask2_ = tap({u + "mask.2"});
  gamma_off_ = pack_vector(arena_, sd_.get(u + "aggregator.gamma").data);
  arena_.upload();
  resolve_encoder(arena_, fnet_);
  resolve_encoder(arena_, cnet_);
  for (PackedConv* L : {&convc1_, &convc2_, &convf1_, &convf2_, &convm_, &to_v_, &to_qk_, &gru_zr_[0], &gru_zr_[1],
                        &gru_q_[0], &gru_q_[1], &fh1_, &fh2_, &mask0_, &mask2_})
    resolve(arena_, *L);
  if (sf) for (PackedConv* L : {&gru_zr_ctx_[0], &gru_zr_ctx_[1], &gru_q_ctx_[0], &gru_q_ctx_[1]}) resolve(arena_, *L);
  gamma_ = arena_.dev(gamma_off_);

  // ---- workspace (sized for maxB pairs; everything stays resident in HBM between calls)
  const int B = maxB;
  const int H2 = conv_out(H, 7, 2, 3), W2 = conv_out(W, 7, 2, 3);
  const long n8 = (long)B * N;
  img4_.alloc(2L * B * H * W * 4);
  for (int i = 0; i < (sf ? 4 : 3); ++i) enc_[i].alloc(2L * B * H2 * W2 * 64);
  fmap_.alloc(2L * B * N * 256);
  const long groups = (long)cdiv(H2 * W2, 64) * 4 + 8;
  psum_.alloc(2L * B * groups * 128); pm2_.alloc(2L * B * groups * 128); pcnt_.alloc(2L * B * groups); fin_.alloc(2L * B * 32 * 128 * 4 * 2);
  for (int i = 0; i < 3; ++i) { mean_[i].alloc(2L * B * 128); rstd_[i].alloc(2L * B * 128); }
  pyrH_[0] = H8; pyrW_[0] = W8;
  for (int l = 1; l < 4; ++l) { pyrH_[l] = pyrH_[l - 1] / 2; pyrW_[l] = pyrW_[l - 1] / 2; }
  ATDN_CHECK(pyrH_[3] >= 2 && pyrW_[3] >= 2, "frame too small for a 4-level pyramid");
  for (int l = 0; l < 4; ++l) {
    brickBW_[l] = cdiv(pyrW_[l], 8); brickBH_[l] = cdiv(pyrH_[l], 4); brickNB_[l] = brickBW_[l] * brickBH_[l] * 32;
    // (bricked: a pair's region holds whole 128-pixel strips, kernels.h: BrickPyramid)
    pyr_[l].alloc(classic_ ? n8 * pyrH_[l] * pyrW_[l] : (long)B * brick_pixel_blocks(N) * kBrickPixelBlock * brickNB_[l]);
  }
  if (sf) {
    for (int l = 0; l < 4; ++l) fbrick_[l].alloc((long)B * brickNB_[l] * 256);
    for (int l = 1; l < 4; ++l) fplain_[l - 1].alloc((long)B * pyrH_[l] * pyrW_[l] * 256);
    coords_used_.alloc(n8 * 2);
  }
  h_[0].alloc(n8 * 128); h_[1].alloc(n8 * 128); x_.alloc(n8 * XLD);
  const AttnGeom ag = attn_geom(B, N, ldN);
  qk_.alloc(n8 * 256); attn_.alloc(std::max(n8 * ldN, attn_floats(ag))); vT_.alloc((long)B * 128 * ldN);
  if (sf) { rowmax_.alloc((long)B * ag.Npad); rinv_.alloc((long)B * ag.Npad); }
  // (only where the split can engage: attn_v_splits() is 1 from 8 pairs per launch on at KITTI size)
  if (sf && low_latency_ && attn_v_splits(attn_geom(1, N, ldN)) > 1) attn_part_.alloc(8L * B * ag.Npad * 128);
  if (sf && low_latency_) {
    for (int i = 0; i < 4; ++i) enc2_[i].alloc((long)B * H2 * W2 * 64);
    ATDN_HIP(hipStreamCreateWithFlags(&par_stream_, hipStreamNonBlocking));
    par_events_.resize(2 * (1 + 64));
    for (auto& e : par_events_) ATDN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  corrfeat_.alloc(n8 * CORR_LD); cor1_.alloc(n8 * 256); corflo_.alloc(n8 * 256); flo1_.alloc(n8 * 128);
  z_.alloc(n8 * 128); rh_.alloc(n8 * 128); fh_.alloc(n8 * 256); mask_.alloc(n8 * 576);
  if (sf) fhG_.alloc(2 * n8 * 18);   // (two copies: one per 128-channel block of the fused flow head)
  coords1_.alloc(n8 * 2); flow4_.alloc(n8 * 4);
  if (sf) for (int p = 0; p < 2; ++p) { pre_zr_[p].alloc(n8 * 256); pre_q_[p].alloc(n8 * 128); }
  // pad lanes that kernels read but never write must be finite zeros
  ATDN_HIP(hipMemset(corrfeat_.p, 0, corrfeat_.n * sizeof(float)));
  ATDN_HIP(hipMemset(vT_.p, 0, vT_.n * sizeof(float)));
  ATDN_HIP(hipMemset(attn_.p, 0, attn_.n * sizeof(float)));
  ATDN_HIP(hipMemset(flow4_.p, 0, flow4_.n * sizeof(float)));
  ATDN_HIP(hipDeviceSynchronize());
  ws_bytes_ = 0;
  DeviceBuf* all[] = {&img4_, &enc_[0], &enc_[1], &enc_[2], &enc_[3], &fmap_, &psum_, &pm2_, &pyr_[0], &pyr_[1], &pyr_[2],
                      &pyr_[3], &h_[0], &h_[1], &x_, &qk_, &attn_, &vT_, &corrfeat_, &cor1_, &corflo_, &flo1_, &z_,
                      &rh_, &fh_, &mask_, &coords1_, &flow4_};
  for (auto* b : all) ws_bytes_ += (size_t)b->n * sizeof(float);
  (void)zero_line();  // allocate the shared zero line now: never inside a stream capture
  sf_counter_attach(); // saturation counter of the sf format (sf.h): attached before the first launch
  ATDN_HIP(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
  ready_ = true;
}

// BasicEncoder.forward (extractor.py:165-189). instance=true: InstanceNorm (fnet); false: BatchNorm folded (cnet).
void GmaNet::run_encoder(const EncoderWeights& E, bool instance, int nimg, hipStream_t st, float** out_buf, int* outH,
                         int* outW) {
  int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
  float* P = enc_[0].p; float* Q = enc_[1].p; float* R = enc_[2].p;
  auto stats_conv = [&](auto mode_tag, const PackedConv& L, const float* src, int ld, int ih, int iw, int stride,
                        int pad, float* dst, int slot) {
    constexpr int MODE = decltype(mode_tag)::value;
    ConvShape s = conv_shape(L, src, ld, (long)ih * iw * ld, nimg, ih, iw, stride, pad, pad);
    const int oh = conv_out(ih, L.KH, stride, pad), ow = conv_out(iw, L.KW, stride, pad);
    EpiBiasStats ep{L.b, dst, (long)oh * ow * L.N, L.N, psum_.p, pm2_.p, 0};
    TileChoice t = conv_dispatch<MODE>(s, ep, st);
    const int groups = cdiv(oh * ow, t.BM) * (t.BM / 32);
    ATDN_CHECK((long)nimg * groups * L.N <= psum_.n, "statistics scratch too small");
    launch_in_finalize_cnt(psum_.p, pm2_.p, nullptr, nimg, groups, oh * ow, L.N, 1e-5f, mean_[slot].p, rstd_[slot].p,
                           reinterpret_cast<double*>(fin_.p), st);
  };
  using TapT = std::integral_constant<int, MODE_TAP>;
  using RowT = std::integral_constant<int, MODE_ROW>;

  // stem: conv 7x7/2 + norm + relu
  if (instance) {
    stats_conv(RowT{}, E.stem, img4_.p, 4, H, W, 2, 3, P, 0);
    launch_in_apply(P, mean_[0].p, rstd_[0].p, nullptr, nullptr, nullptr, nimg, (long)h * w, 64, st);
  } else {
    ConvShape s = conv_shape(E.stem, img4_.p, 4, (long)H * W * 4, nimg, H, W, 2, 3, 3);
    conv_dispatch<MODE_ROW>(s, EpiBias<ACT_RELU>{E.stem.b, P, (long)h * w * 64, 64, 1.f}, st);
  }
  int c = 64;
  for (int bi = 0; bi < 6; ++bi) {
    const auto& Bk = E.blk[bi];
    const int stride = Bk.has_ds ? 2 : 1;
    const int co = Bk.c1.N;
    const int oh = conv_out(h, 3, stride, 1), ow = conv_out(w, 3, stride, 1);
    const long ohw = (long)oh * ow;
    if (instance) {
      stats_conv(TapT{}, Bk.c1, P, c, h, w, stride, 1, Q, 0);
      launch_in_apply(Q, mean_[0].p, rstd_[0].p, nullptr, nullptr, nullptr, nimg, ohw, co, st);
      stats_conv(TapT{}, Bk.c2, Q, co, oh, ow, 1, 1, R, 0);
      if (Bk.has_ds) {
        stats_conv(TapT{}, Bk.ds, P, c, h, w, 2, 0, Q, 1);
        launch_in_apply(R, mean_[0].p, rstd_[0].p, Q, mean_[1].p, rstd_[1].p, nimg, ohw, co, st);
      } else {
        launch_in_apply(R, mean_[0].p, rstd_[0].p, P, nullptr, nullptr, nimg, ohw, co, st);
      }
      std::swap(P, R);  // block output becomes the next input; Q, R are free again
    } else {
      ConvShape s1 = conv_shape(Bk.c1, P, c, (long)h * w * c, nimg, h, w, stride, 1, 1);
      conv_dispatch<MODE_TAP>(s1, EpiBias<ACT_RELU>{Bk.c1.b, Q, ohw * co, co, 1.f}, st);
      ConvShape s2 = conv_shape(Bk.c2, Q, co, ohw * co, nimg, oh, ow, 1, 1, 1);
      if (Bk.has_ds) {
        ConvShape sd = conv_shape(Bk.ds, P, c, (long)h * w * c, nimg, h, w, 2, 0, 0);
        conv_dispatch<MODE_TAP>(sd, EpiBias<ACT_NONE>{Bk.ds.b, R, ohw * co, co, 1.f}, st);
        // P (the block input) is dead once the downsample conv has read it: reuse it for the output
        conv_dispatch<MODE_TAP>(s2, EpiBiasReluAddRelu{Bk.c2.b, R, ohw * co, co, P, ohw * co, co}, st);
      } else {
        conv_dispatch<MODE_TAP>(s2, EpiBiasReluAddRelu{Bk.c2.b, P, ohw * co, co, R, ohw * co, co}, st);
        std::swap(P, R);
      }
    }
    h = oh; w = ow; c = co;
  }
  *out_buf = P; *outH = h; *outW = w;
}

void GmaNet::iteration(int B, hipStream_t st) {
  const long n8 = (long)B * N;
  // -- index the correlation pyramid at the current coordinates (corr.py:32-53)
  PyramidLevels pl;
  for (int l = 0; l < 4; ++l) { pl.base[l] = pyr_[l].p; pl.H[l] = pyrH_[l]; pl.W[l] = pyrW_[l]; }
  launch_lookup(pl, coords1_.p, n8, corrfeat_.p, CORR_LD, st);
  mark(ST_LOOKUP, st);

  // -- motion encoder (update.py:76-84); torch.cat is realised by writing channel slices
  ConvShape s = conv_shape(convc1_, corrfeat_.p, CORR_LD, (long)N * CORR_LD, B, H8, W8, 1, 0, 0);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{convc1_.b, cor1_.p, (long)N * 256, 256, 1.f}, st);
  mark(ST_CONVC1, st);
  s = conv_shape(convc2_, cor1_.p, 256, (long)N * 256, B, H8, W8, 1, 1, 1);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{convc2_.b, corflo_.p, (long)N * 256, 256, 1.f}, st);
  s = conv_shape(convf1_, flow4_.p, 4, (long)N * 4, B, H8, W8, 1, 3, 3);
  conv_dispatch<MODE_ROW>(s, EpiBias<ACT_RELU>{convf1_.b, flo1_.p, (long)N * 128, 128, 1.f}, st);
  s = conv_shape(convf2_, flo1_.p, 128, (long)N * 128, B, H8, W8, 1, 1, 1);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{convf2_.b, corflo_.p + 192, (long)N * 256, 256, 1.f}, st);
  s = conv_shape(convm_, corflo_.p, 256, (long)N * 256, B, H8, W8, 1, 1, 1);
  float* mf = x_.p + 128;  // motion_features: 126 conv channels + 2 flow channels (written by the flow update)
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{convm_.b, mf, (long)N * XLD, XLD, 1.f}, st);
  mark(ST_MOTION, st);

  // -- global motion aggregation (gma.py:102-115): v^T, then attn @ v with the residual fused
  s = conv_shape(to_v_, mf, XLD, (long)N * XLD, B, H8, W8, 1, 0, 0);
  conv_dispatch<MODE_TAP>(s, EpiStoreT{vT_.p, (long)128 * ldN, ldN}, st);
  mark(ST_AGG_VT, st);
  ConvShape a;
  a.src0 = attn_.p; a.ld0 = ldN; a.sb0 = (long)N * ldN; a.C0 = ldN; a.H = 1; a.W = N;
  a.w = vT_.p; a.wb = (long)128 * ldN; a.ldw = ldN; a.N = 128; a.nimg = B;
  conv_dispatch<MODE_TAP>(a, EpiAggregate{gamma_, mf, (long)N * XLD, XLD, x_.p + 256, (long)N * XLD, XLD}, st);
  mark(ST_AGG, st);

  // -- separable ConvGRU (update.py:48-63): horizontal (1x5) then vertical (5x1)
  for (int p = 0; p < 2; ++p) {
    const float* hin = h_[p].p;
    float* hout = h_[p ^ 1].p;
    const int ph = p ? 2 : 0, pw = p ? 0 : 2;
    ConvShape g = conv_shape(gru_zr_[p], hin, 128, (long)N * 128, B, H8, W8, 1, ph, pw);
    g.C0 = 128; g.src1 = x_.p; g.ld1 = XLD; g.sb1 = (long)N * XLD; g.C1 = XLD;
    conv_dispatch<MODE_TAP>(g, EpiGruZR{gru_zr_[p].b, hin, z_.p, rh_.p, (long)N * 128}, st);
    mark(p ? ST_GRU_ZR_V : ST_GRU_ZR, st);
    g.src0 = rh_.p; g.w = gru_q_[p].w; g.ldw = gru_q_[p].ldw; g.N = gru_q_[p].N;
    conv_dispatch<MODE_TAP>(g, EpiGruQ{gru_q_[p].b, hin, z_.p, hout, (long)N * 128}, st);
    mark(p ? ST_GRU_Q_V : ST_GRU_Q, st);
  }
  // two passes: the state is back in h_[0]

  // -- flow head (update.py:7-15) and coordinate update (network.py:116)
  s = conv_shape(fh1_, h_[0].p, 128, (long)N * 128, B, H8, W8, 1, 1, 1);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{fh1_.b, fh_.p, (long)N * 256, 256, 1.f}, st);
  s = conv_shape(fh2_, fh_.p, 256, (long)N * 256, B, H8, W8, 1, 1, 1);
  conv_dispatch<MODE_TAP>(s, EpiFlowDelta{fh2_.b, coords1_.p, flow4_.p, x_.p + 254, XLD, (long)N * XLD, W8, (long)N}, st);
  mark(ST_FLOWHEAD, st);
}

void GmaNet::run_body(int B, int iters, hipStream_t st) {
  // ---- feature network on [im1 batch ‖ im2 batch] (network.py:86)
  float* f; int fh, fw;
  run_encoder(fnet_, true, 2 * B, st, &f, &fh, &fw);
  ATDN_CHECK(fh == H8 && fw == W8, "encoder geometry mismatch");
  ConvShape s = conv_shape(fnet_.head, f, 128, (long)N * 128, 2 * B, H8, W8, 1, 0, 0);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_NONE>{fnet_.head.b, fmap_.p, (long)N * 256, 256, 1.f}, st);
  mark(ST_FNET, st);

  // ---- all-pairs correlation (corr.py:55-63) and its 4-level pyramid (corr.py:28-30)
  ConvShape c;
  c.src0 = fmap_.p; c.ld0 = 256; c.sb0 = (long)N * 256; c.C0 = 256; c.H = 1; c.W = N;
  c.w = fmap_.p + (long)B * N * 256; c.wb = (long)N * 256; c.ldw = 256; c.N = N; c.nimg = B;
  conv_dispatch<MODE_TAP>(c, EpiScale{1.0f / sqrtf(256.0f), pyr_[0].p, (long)N * N, N}, st);
  mark(ST_CORR, st);
  for (int l = 1; l < 4; ++l) launch_avgpool(pyr_[l - 1].p, pyrH_[l - 1], pyrW_[l - 1], pyr_[l].p, (long)B * N, st);
  mark(ST_POOL, st);

  // ---- context network on im1 (network.py:94-97): tanh -> hidden state, relu -> x[:, 0:128]
  run_encoder(cnet_, false, B, st, &f, &fh, &fw);
  s = conv_shape(cnet_.head, f, 128, (long)N * 128, B, H8, W8, 1, 0, 0);
  conv_dispatch<MODE_TAP>(s, EpiContextSplit{cnet_.head.b, h_[0].p, (long)N * 128, x_.p, (long)N * XLD, XLD}, st);
  mark(ST_CNET, st);

  // ---- attention (gma.py:54-76): q,k projection, q·k^T, row softmax
  s = conv_shape(to_qk_, x_.p, XLD, (long)N * XLD, B, H8, W8, 1, 0, 0);
  conv_dispatch<MODE_TAP>(s, EpiQK{1.0f / sqrtf(128.0f), 128, qk_.p, (long)N * 256, 256}, st);
  ConvShape q;
  q.src0 = qk_.p; q.ld0 = 256; q.sb0 = (long)N * 256; q.C0 = 128; q.H = 1; q.W = N;
  q.w = qk_.p + 128; q.wb = (long)N * 256; q.ldw = 256; q.N = N; q.nimg = B;
  conv_dispatch<MODE_TAP>(q, EpiScale{1.0f, attn_.p, (long)N * ldN, ldN}, st);
  mark(ST_ATTN_LOGITS, st);
  launch_softmax_rows(attn_.p, (long)B * N, N, ldN, st);
  mark(ST_ATTN, st);

  for (int it = 0; it < iters; ++it) {
    iteration(B, st);
    if (preds_out_) {   // forward_predictions: every iteration's flow through its own mask (network.py:118-124)
      mask_head(B, st);
      launch_upsample(mask_.p, flow4_.p, B, H8, W8, nullptr, preds_out_ + it * preds_stride_, st);
    }
  }
  // ---- mask head, once (update.py:120-123,138): only the last iteration's mask reaches the output
  if (!preds_out_) mask_head(B, st);
  mark(ST_MASK, st);
}

void GmaNet::mask_head(int B, hipStream_t st) {
  ConvShape s = conv_shape(mask0_, h_[0].p, 128, (long)N * 128, B, H8, W8, 1, 1, 1);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{mask0_.b, fh_.p, (long)N * 256, 256, 1.f}, st);
  s = conv_shape(mask2_, fh_.p, 256, (long)N * 256, B, H8, W8, 1, 0, 0);
  conv_dispatch<MODE_TAP>(s, EpiBias<ACT_NONE>{mask2_.b, mask_.p, (long)N * 576, 576, 0.25f}, st);
}

// =============================================================== split-f16 pipeline (precision == 1)
// Same op sequence; every TAP-mode GEMM runs on conv_sf_kernel and every tensor that feeds one is stored in the
// sf format (sf.h). ROW-mode layers (7x7 stems, convf1) stay on the exact-fp32 engine and write sf directly.
void GmaNet::run_encoder_sf(const EncoderWeights& E, bool instance, int nimg, hipStream_t st, float** out_buf, int first_img,
                            DeviceBuf* bufs) {
  const float* images = img4_.p + (long)first_img * H * W * 4;
  int h = conv_out(H, 7, 2, 3), w = conv_out(W, 7, 2, 3);
  DeviceBuf* eb = bufs ? bufs : enc_;
  float* X = eb[0].p; float* R = eb[1].p; float* Y = eb[2].p; float* O = eb[3].p;
  // in_slot >= 0: `src` is the RAW output of the previous statistics conv and mean_/rstd_[in_slot] are its statistics
  // (normalise-on-load: the conv's patch loader applies InstanceNorm + ReLU itself)
  auto stats_sf = [&](const PackedConv& L, const float* src, int ld, int ih, int iw, int stride, int pad, float* dst,
                      int slot, int in_slot = -1) {
    ConvShape s = conv_shape(L, src, ld, (long)ih * iw * ld, nimg, ih, iw, stride, pad, pad);
    if (in_slot >= 0) { s.in_mean = mean_[in_slot].p; s.in_rstd = rstd_[in_slot].p; }
    const int oh = conv_out(ih, L.KH, stride, pad), ow = conv_out(iw, L.KW, stride, pad);
    EpiBiasStats ep{L.b, dst, (long)oh * ow * L.N, L.N, psum_.p, pm2_.p, 0, pcnt_.p};
    TileChoice t = conv_sf_dispatch(s, L.wscale, ep, st);
    const int groups = t.groups_per_img;
    ATDN_CHECK((long)nimg * groups * L.N <= psum_.n && (long)nimg * groups <= pcnt_.n, "statistics scratch too small");
    if (t.counted)
      launch_in_finalize_cnt(psum_.p, pm2_.p, pcnt_.p, nimg, groups, oh * ow, L.N, 1e-5f, mean_[slot].p,
                             rstd_[slot].p, reinterpret_cast<double*>(fin_.p), st);
    else  // 1-D M tiling: the valid rows of a group follow from its index (part_cnt = nullptr)
      launch_in_finalize_cnt(psum_.p, pm2_.p, nullptr, nimg, groups, oh * ow, L.N, 1e-5f, mean_[slot].p, rstd_[slot].p,
                             reinterpret_cast<double*>(fin_.p), st);
  };
  // 7x7 stem on the split-f16 engine (stem_sf.hip); with InstanceNorm the conv runs twice (statistics, then
  // normalise + ReLU -> sf) instead of materialising the raw tensor
  // (round 5, split-f16 mode: ONE pass — raw fp32 output + statistics; the first block normalises it on load, below)
  const bool stem_raw = instance && norm_on_load_;
  if (stem_raw) {
    const int groups = stem_sf_groups(h, w);
    ATDN_CHECK((long)nimg * groups * 64 <= psum_.n && (long)nimg * groups <= pcnt_.n, "statistics scratch too small");
    launch_stem_sf(3, images, nimg, H, W, E.stem.wf, E.stem.wscale, E.stem.b, X, psum_.p, pm2_.p, pcnt_.p, nullptr, nullptr, st);
    launch_in_finalize_cnt(psum_.p, pm2_.p, pcnt_.p, nimg, groups, h * w, 64, 1e-5f, mean_[2].p, rstd_[2].p,
                           reinterpret_cast<double*>(fin_.p), st);
  } else if (instance) {
    const int groups = stem_sf_groups(h, w);
    ATDN_CHECK((long)nimg * groups * 64 <= psum_.n && (long)nimg * groups <= pcnt_.n, "statistics scratch too small");
    launch_stem_sf(1, images, nimg, H, W, E.stem.wf, E.stem.wscale, E.stem.b, nullptr, psum_.p, pm2_.p, pcnt_.p, nullptr,
                   nullptr, st);
    launch_in_finalize_cnt(psum_.p, pm2_.p, pcnt_.p, nimg, groups, h * w, 64, 1e-5f, mean_[0].p, rstd_[0].p,
                           reinterpret_cast<double*>(fin_.p), st);
    launch_stem_sf(2, images, nimg, H, W, E.stem.wf, E.stem.wscale, E.stem.b, X, nullptr, nullptr, nullptr, mean_[0].p,
                   rstd_[0].p, st);
  } else {
    launch_stem_sf(0, images, nimg, H, W, E.stem.wf, E.stem.wscale, E.stem.b, X, nullptr, nullptr, nullptr, nullptr,
                   nullptr, st);
  }
  int c = 64;
  for (int bi = 0; bi < 6; ++bi) {
    const auto& Bk = E.blk[bi];
    const int stride = Bk.has_ds ? 2 : 1;
    const int co = Bk.c1.N;
    const int oh = conv_out(h, 3, stride, 1), ow = conv_out(w, 3, stride, 1);
    const long ohw = (long)oh * ow;
    if (instance && norm_on_load_) {
      // conv1 -> (InstanceNorm + ReLU inside conv2's patch loader) -> conv2: the pass that used to materialise
      // relu(IN(conv1)) between them (4 B read + 4 B written per element) is gone. conv2's statistics land in slot 1
      // because slot 0 (conv1's) is read by conv2 itself.
      const bool raw_in = stem_raw && bi == 0;   // X is the stem's RAW output: conv1 normalises on load, the shortcut on the fly
      ATDN_CHECK(!raw_in || !Bk.has_ds, "the block behind the stem has no downsample branch");
      stats_sf(Bk.c1, X, c, h, w, stride, 1, R, 0, raw_in ? 2 : -1);
      stats_sf(Bk.c2, R, co, oh, ow, 1, 1, Y, 1, 0);
      if (Bk.has_ds) {
        stats_sf(Bk.ds, X, c, h, w, 2, 0, R, 0);  // R is dead once conv2 has consumed it: holds the raw shortcut
        launch_in_apply_sf(Y, O, mean_[1].p, rstd_[1].p, nullptr, R, mean_[0].p, rstd_[0].p, nimg, ohw, co, st);
      } else if (raw_in) {
        launch_in_apply_sf(Y, O, mean_[1].p, rstd_[1].p, nullptr, X, mean_[2].p, rstd_[2].p, nimg, ohw, co, st, true);
      } else {
        launch_in_apply_sf(Y, O, mean_[1].p, rstd_[1].p, X, nullptr, nullptr, nullptr, nimg, ohw, co, st);
      }
    } else if (instance) {
      stats_sf(Bk.c1, X, c, h, w, stride, 1, R, 0);
      launch_in_apply_sf(R, Y, mean_[0].p, rstd_[0].p, nullptr, nullptr, nullptr, nullptr, nimg, ohw, co, st);
      stats_sf(Bk.c2, Y, co, oh, ow, 1, 1, R, 0);
      if (Bk.has_ds) {
        stats_sf(Bk.ds, X, c, h, w, 2, 0, Y, 1);  // Y is dead once conv2 has consumed it: holds the raw shortcut
        launch_in_apply_sf(R, O, mean_[0].p, rstd_[0].p, nullptr, Y, mean_[1].p, rstd_[1].p, nimg, ohw, co, st);
      } else {
        launch_in_apply_sf(R, O, mean_[0].p, rstd_[0].p, X, nullptr, nullptr, nullptr, nimg, ohw, co, st);
      }
    } else {
      const float* res = X;
      if (Bk.has_ds) {
        ConvShape sd = conv_shape(Bk.ds, X, c, (long)h * w * c, nimg, h, w, 2, 0, 0);
        conv_sf_dispatch(sd, Bk.ds.wscale, SfBias<ACT_NONE>{Bk.ds.b, R, ohw * co, co}, st);
        res = R;
      }
      ConvShape s1 = conv_shape(Bk.c1, X, c, (long)h * w * c, nimg, h, w, stride, 1, 1);
      conv_sf_dispatch(s1, Bk.c1.wscale, SfBias<ACT_RELU>{Bk.c1.b, Y, ohw * co, co}, st);
      ConvShape s2 = conv_shape(Bk.c2, Y, co, ohw * co, nimg, oh, ow, 1, 1, 1);
      conv_sf_dispatch(s2, Bk.c2.wscale, SfBiasReluAddRelu{Bk.c2.b, res, ohw * co, co, O, ohw * co, co}, st);
    }
    std::swap(X, O);
    h = oh; w = ow; c = co;
  }
  ATDN_CHECK(h == H8 && w == W8, "encoder geometry mismatch");
  *out_buf = X;
}

void GmaNet::iteration_sf(int B, hipStream_t st) {
  const long n8 = (long)B * N;
  ConvShape s;
  if (par_) fork(st, par_stream_);   // the flow branch starts where the previous iteration (or the set-up) ended
  // cor1 = relu(convc1(lookup(coords1))) in one kernel: the 324 samples of a pixel never leave the CU
  launch_lookup_conv(brick_pyramid(), coords1_.p, n8, coords_used_.p, convc1_.wf16, convc1_.wscale, convc1_.b, cor1_.p,
                     sf_fast_mode(), st);
  mark(ST_LOOKUP, st);
  // (Round 3 captured the flow branch — convf1, convf2: independent of the correlation branch — as a parallel branch of the
  // graph on a second stream: +0.3 % on bench.py's two-stream loop, but -3 % in the sequence driver, whose lane streams then
  // share hardware queues with the branch streams of the two graphs. Removed: one stream per clip, one queue per stream.
  // Round 5: the two 3x3 convolutions of the two branches, convc2 and convf2, run the same kernel on different operands — they go
  // out as ONE launch (conv_sf6_pair_kernel): convf2's 640 blocks fill the tail of convc2's 1,920 (512 resident: 3.75 rounds +
  // 1.25 rounds become 5) — motion encoder 5.80 -> 5.57 ms per forward on ONE stream; under bench.py's two streams the other
  // clip's launches were already filling those tails and the rate does not move (410.3 / 410.3 / 408.9 against 410.5 / 410.3 /
  // 409.1 pairs/s, profiles/r05_ab_pair_launch.txt). Kept for callers with one stream.)
  s = conv_shape(convc2_, cor1_.p, 256, (long)N * 256, B, H8, W8, 1, 1, 1);
  const ConvShape sf2 = conv_shape(convf2_, flo1_.p, 128, (long)N * 128, B, H8, W8, 1, 1, 1);
  if (par_) {
    // low-latency capture: the flow branch (convf1 -> convf2) runs beside the correlation branch (lookup + convc1 -> convc2);
    // it was forked at the top of the iteration (the lookup above is already on `st`)
    launch_flow_conv7_sf(flow4_.p, B, H8, W8, arena_.dev(convf1_sf_off_), convf1_wscale_, convf1_.b, flo1_.p, sf_fast_mode(), par_stream_);
    conv_sf_dispatch(sf2, convf2_.wscale, SfBias<ACT_RELU>{convf2_.b, corflo_.p + 192, (long)N * 256, 256}, par_stream_);
    conv_sf_dispatch(s, convc2_.wscale, SfBias<ACT_RELU>{convc2_.b, corflo_.p, (long)N * 256, 256}, st);
    fork(par_stream_, st);   // join
  } else {
    launch_flow_conv7_sf(flow4_.p, B, H8, W8, arena_.dev(convf1_sf_off_), convf1_wscale_, convf1_.b, flo1_.p, sf_fast_mode(), st);
    conv_sf_dispatch_pair(s, convc2_.wscale, SfBias<ACT_RELU>{convc2_.b, corflo_.p, (long)N * 256, 256},
                          sf2, convf2_.wscale, SfBias<ACT_RELU>{convf2_.b, corflo_.p + 192, (long)N * 256, 256}, st);
  }
  s = conv_shape(convm_, corflo_.p, 256, (long)N * 256, B, H8, W8, 1, 1, 1);
  float* mf = x_.p + 128;
  conv_sf_dispatch(s, convm_.wscale, SfBias<ACT_RELU>{convm_.b, mf, (long)N * XLD, XLD}, st);
  mark(ST_MOTION, st);

  // v^T [128][pixels] directly: the projection matrix is the A operand (128 rows), the motion features are the
  // K-contiguous "weight" rows, so the sf store runs along pixels (no transposed 2-byte scatter)
  ConvShape v;
  v.src0 = to_v_.w; v.ld0 = 128; v.sb0 = 0; v.C0 = 128; v.H = 1; v.W = 128;
  v.w = mf; v.wb = (long)N * XLD; v.ldw = XLD; v.N = N; v.nimg = B;
  conv_sf_dispatch(v, to_v_.wscale, SfVT{vT_.p, (long)128 * ldN, ldN}, st);   // keys of a chunk in operand order
  mark(ST_AGG_VT, st);
  launch_attn_v(attn_.p, rinv_.p, attn_geom(B, N, ldN), vT_.p, gamma_, mf, x_.p + 256, (long)N * XLD, XLD,
                sf_fast_mode(), st, attn_part_.p);   // (attn_part_ is null unless the handle was built low-latency)
  mark(ST_AGG, st);

  for (int p = 0; p < 2; ++p) {
    const float* hin = h_[p].p;
    float* hout = h_[p ^ 1].p;
    const int ph = p ? 2 : 0, pw = p ? 0 : 2;
    ConvShape g = conv_shape(gru_zr_[p], hin, 128, (long)N * 128, B, H8, W8, 1, ph, pw);
    g.C0 = 128; g.src1 = x_.p + 128; g.ld1 = XLD; g.sb1 = (long)N * XLD; g.C1 = 256;  // [h | motion | motion_global]
    conv_sf_dispatch(g, gru_zr_[p].wscale,
                     SfGruZR{gru_zr_[p].b, hin, z_.p, rh_.p, (long)N * 128, pre_zr_[p].p, (long)N * 256}, st);
    mark(p ? ST_GRU_ZR_V : ST_GRU_ZR, st);
    g.src0 = rh_.p; g.w = gru_q_[p].w; g.wfrag16 = gru_q_[p].wf16; g.ldw = gru_q_[p].ldw; g.N = gru_q_[p].N;
    conv_sf_dispatch(g, gru_q_[p].wscale, SfGruQ{gru_q_[p].b, hin, z_.p, hout, (long)N * 128, pre_q_[p].p}, st);
    mark(p ? ST_GRU_Q_V : ST_GRU_Q, st);
  }

  s = conv_shape(fh1_, h_[0].p, 128, (long)N * 128, B, H8, W8, 1, 1, 1);
  const SfFlowDelta fd{fh2_.b, coords1_.p, flow4_.p, x_.p, XLD, (long)N * XLD, 254, W8, (long)N};
  // conv1 with conv2's partial sums in its epilogue, then the 3 x 3 gather (small_convs.h)
  const long fh_gs = (long)maxB * N * 18;   // second copy of the partial sums: channels 128..255 (conv_sf_inst_e.hip)
  launch_flow_head_fused(s, fh1_.wscale, SfFlowHeadPartial{fh1_.b, arena_.dev(fh2_w32_off_), fhG_.p, (long)N, fh2_mul_, 1.0f / fh2_mul_, fh_gs}, st);
  launch_flow_gather(fhG_.p, fh_gs, B, H8, W8, fd, st);
  mark(ST_FLOWHEAD, st);
}

void GmaNet::run_body_sf(int B, int iters, hipStream_t st) {
  float* f;
  // feature-network passes: two per pair; one per frame in sequence mode; in a continued sequence frame 0's features
  // were already moved into fmap_ slot 0 by forward_sequence and only frames 1..B are encoded (img4_ still holds all
  // B+1 frames: the context network needs frame 0)
  const int nfeat = seq_ == 0 ? 2 * B : seq_ == 1 ? B + 1 : B;
  // low-latency capture (par_): the context chain (context network -> attention -> GRU context terms) is a branch of its own
  // beside the feature chain (feature network -> correlation pyramid); they share no tensor (enc2_: the context network's maps)
  hipStream_t sB = par_ ? par_stream_ : st;
  if (par_) fork(st, sB);
  // (Round 4 measured the encoders depth-first in sub-batches of <= 2-8 frames, so that a conv's output and its consumer's
  // input fit the 256 MiB Infinity Cache together: every extra launch cost 10-12 us and nothing came back from the cache,
  // profiles/r04_ab_encoder_subbatch.txt. All frames of a launch go through a layer together.)
  run_encoder_sf(fnet_, true, nfeat, st, &f, seq_ == 2 ? 1 : 0);
  ConvShape s = conv_shape(fnet_.head, f, 128, (long)N * 128, nfeat, H8, W8, 1, 0, 0);
  float* fdst = fmap_.p + (seq_ == 2 ? (long)N * 256 : 0);
  conv_sf_dispatch(s, fnet_.head.wscale, SfBias<ACT_NONE>{fnet_.head.b, fdst, (long)N * 256, 256}, st);
  mark(ST_FNET, st);

  // all-pairs correlation (corr.py:55-63) and its pyramid (corr.py:16-30): every level = fmap1 x (target features of that level
  // in BRICK order)^T, written by corr_bricks_kernel in the brick-major layout the lookup reads; padding cells are zero feature
  // rows. Levels 1-3: 2x2-pooled features — correlation is linear in the target features, so avg_pool2d of corr.py:28-30
  // commutes with the dot product. fmap_ slot b is frame b (sequence modes: pair b = frames b, b + 1) or [im1 batch | im2 batch].
  const float* target = fmap_.p + (long)(seq_ ? 1 : B) * N * 256;
  for (int l = 0; l < 4; ++l) {
    const float* plain = target;
    long plain_sb = (long)N * 256;
    if (l > 0) {
      const float* prev = l == 1 ? target : fplain_[l - 2].p;
      const long prev_sb = l == 1 ? (long)N * 256 : (long)pyrH_[l - 1] * pyrW_[l - 1] * 256;
      plain_sb = (long)pyrH_[l] * pyrW_[l] * 256;
      launch_pool_features_sf(prev, B, pyrH_[l - 1], pyrW_[l - 1], 256, prev_sb, fplain_[l - 1].p, plain_sb, st);
      plain = fplain_[l - 1].p;
    }
    launch_brick_rows(plain, plain_sb, B, pyrH_[l], pyrW_[l], 256, fbrick_[l].p, (long)brickNB_[l] * 256, st);
    launch_corr_bricks(fmap_.p, (long)N * 256, fbrick_[l].p, (long)brickNB_[l] * 256, B, N, brickNB_[l], 1.0f / sqrtf(256.0f), pyr_[l].p,
                       sf_fast_mode(), st);
    if (l == 0) mark(ST_CORR, st);
  }
  mark(ST_POOL, st);

  run_encoder_sf(cnet_, false, B, sB, &f, 0, par_ ? enc2_ : nullptr);
  s = conv_shape(cnet_.head, f, 128, (long)N * 128, B, H8, W8, 1, 0, 0);
  conv_sf_dispatch(s, cnet_.head.wscale,
                   SfContextSplit{cnet_.head.b, h_[0].p, (long)N * 128, x_.p, (long)N * XLD, XLD}, sB);
  mark(ST_CNET, st);

  s = conv_shape(to_qk_, x_.p, XLD, (long)N * XLD, B, H8, W8, 1, 0, 0);
  conv_sf_dispatch(s, to_qk_.wscale, SfQK{1.0f / sqrtf(128.0f), 128, qk_.p, (long)N * 256, 256}, sB);
  {
    // Q K^T with the row softmax fused in (attention.hip): a cheap first sweep (f16 x f16 logits) for the row maxima,
    // then the full-precision sweep that writes exp(s - max) in MFMA-operand order and the row sums
    const AttnGeom ag = attn_geom(B, N, ldN);
    launch_qk_rowmax(qk_.p, ag, rowmax_.p, sB);
    mark(ST_ATTN_LOGITS, st);
    launch_qk_softmax(qk_.p, ag, rowmax_.p, attn_.p, rinv_.p, sf_fast_mode(), sB);
  }
  mark(ST_ATTN, st);

  // context-channel part of the six ConvGRU convolutions: identical in every iteration, computed once
  for (int p = 0; p < 2; ++p) {
    const int ph = p ? 2 : 0, pw = p ? 0 : 2;
    ConvShape g = conv_shape(gru_zr_ctx_[p], x_.p, XLD, (long)N * XLD, B, H8, W8, 1, ph, pw);
    conv_sf_dispatch(g, gru_zr_ctx_[p].wscale, EpiBias<ACT_NONE>{nullptr, pre_zr_[p].p, (long)N * 256, 256, 1.f}, sB);
    g = conv_shape(gru_q_ctx_[p], x_.p, XLD, (long)N * XLD, B, H8, W8, 1, ph, pw);
    conv_sf_dispatch(g, gru_q_ctx_[p].wscale, EpiBias<ACT_NONE>{nullptr, pre_q_[p].p, (long)N * 128, 128, 1.f}, sB);
  }
  mark(ST_GRU_CTX, st);
  if (par_) fork(sB, st);   // join: the iterations need both chains

  for (int it = 0; it < iters; ++it) {
    iteration_sf(B, st);
    if (preds_out_) {   // forward_predictions: every iteration's flow through its own mask (network.py:118-124)
      mask_head_sf(B, st);
      launch_upsample(mask_.p, flow4_.p, B, H8, W8, nullptr, preds_out_ + it * preds_stride_, st);
    }
  }
  if (!preds_out_) mask_head_sf(B, st);   // test mode: only the last iteration's mask reaches the output (update.py:120-123,138)
  mark(ST_MASK, st);
}

void GmaNet::mask_head_sf(int B, hipStream_t st) {
  ConvShape s = conv_shape(mask0_, h_[0].p, 128, (long)N * 128, B, H8, W8, 1, 1, 1);
  conv_sf_dispatch(s, mask0_.wscale, SfBias<ACT_RELU>{mask0_.b, fh_.p, (long)N * 256, 256}, st);
  s = conv_shape(mask2_, fh_.p, 256, (long)N * 256, B, H8, W8, 1, 0, 0);
  conv_sf_dispatch(s, mask2_.wscale, EpiBias<ACT_NONE>{mask2_.b, mask_.p, (long)N * 576, 576, 0.25f}, st);
}

void GmaNet::capture(int B, int iters) {
  // parallel branches: low-latency handles, one or two pairs per launch (gma.h). If a capture with branches cannot be built (a
  // runtime that refuses the cross-stream dependency), the same body is captured again as one chain: same kernels either way.
  // (ATDN_PAR_FORCE=1: branches at any batch size — the A/B of DESIGN.md section 10.8, nothing else)
  static const bool par_force = getenv("ATDN_PAR_FORCE") && getenv("ATDN_PAR_FORCE")[0] == '1';
  const bool want_par = low_latency_ && precision >= 1 && (B <= 2 || par_force) && par_stream_ != nullptr && !preds_out_ && par_ok_;
  for (int attempt = want_par ? 0 : 1; attempt < 2; ++attempt) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ATDN_HIP(hipStreamBeginCapture(cap_stream_, hipStreamCaptureModeThreadLocal));
    par_ = attempt == 0;
    par_next_ = 0;
    try {
      if (precision >= 1) { FastGuard fg(precision == 2); run_body_sf(B, iters, cap_stream_); } else run_body(B, iters, cap_stream_);
      par_ = false;
      ATDN_HIP(hipStreamEndCapture(cap_stream_, &graph));
      ATDN_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    } catch (...) {
      const bool was_par = par_ || attempt == 0;
      par_ = false;
      hipGraph_t g2 = nullptr;
      (void)hipStreamEndCapture(cap_stream_, &g2);
      if (g2) (void)hipGraphDestroy(g2);
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      if (attempt == 0 && was_par) {
        fprintf(stderr, "atdn: the flow graph could not be captured with parallel branches; capturing it as one chain\n");
        par_ok_ = false;
        continue;
      }
      throw;
    }
    (void)hipGraphDestroy(graph);
    graphs_[{B, iters * 4 + seq_}] = exec;
    return;
  }
}

void GmaNet::forward_sequence(const float* frames, int B, int iters, const float* flow_init, float* flow_low,
                               float* flow_up, hipStream_t st, bool continued) {
  ATDN_CHECK(ready_, "weights not finalized");
  ATDN_CHECK(precision >= 1, "sequence mode is built for the split-f16 pipeline");
  ATDN_CHECK(B >= 1 && B <= maxB, "batch exceeds max_batch of this handle");
  ATDN_CHECK(iters >= 1 && iters <= 64, "iters out of range");
  ATDN_CHECK(frames && flow_low && flow_up, "null tensor");
  const long frame = 3L * H * W;
  if (continued) {
    // frame 0 is the last frame of the previous sequence call on this handle: its features move to slot 0 and the
    // feature network only sees frames 1..B (every frame of a long sequence passes through it exactly once)
    ATDN_CHECK(last_frame_ >= 1, "continued sequence call without a previous sequence call on this handle");
    ATDN_HIP(hipMemcpyAsync(fmap_.p, fmap_.p + (long)last_frame_ * N * 256, (size_t)N * 256 * sizeof(float),
                            hipMemcpyDeviceToDevice, st));
    seq_ = 2;
  } else {
    seq_ = 1;
  }
  // frames 0..B-1 are the first images, frame B the last second image: img4 = [frame 0 .. frame B]
  last_frame_ = 0;   // stays 0 if anything below throws: the next call cannot continue from a half-launched clip
  launch_prep_images(frames, frames + (long)B * frame, B, H, W, img4_.p, st, 1);
  launch_init_coords_sf(flow_init, B, H8, W8, coords1_.p, flow4_.p, x_.p, XLD, 254, st);
  launch_body(B, iters, st);
  launch_upsample(mask_.p, flow4_.p, B, H8, W8, flow_low, flow_up, st);
  last_frame_ = B;
}

void GmaNet::launch_body(int B, int iters, hipStream_t st) {
  last_B_ = B;
  if (use_graph_) {
    auto key = std::make_pair(B, iters * 4 + seq_);
    if (!graphs_.count(key)) capture(B, iters);
    ATDN_HIP(hipGraphLaunch(graphs_[key], st));
  } else {
    if (precision >= 1) { FastGuard fg(precision == 2); run_body_sf(B, iters, st); } else run_body(B, iters, st);
  }
}

void GmaNet::forward(const float* im1, const float* im2, int B, int iters, const float* flow_init, float* flow_low,
                     float* flow_up, hipStream_t st) {
  ATDN_CHECK(ready_, "weights not finalized");
  ATDN_CHECK(B >= 1 && B <= maxB, "batch exceeds max_batch of this handle");
  ATDN_CHECK(iters >= 1 && iters <= 64, "iters out of range");
  ATDN_CHECK(im1 && im2 && flow_low && flow_up, "null tensor");
  seq_ = 0;
  last_frame_ = 0;   // pair mode overwrites fmap_: a later continued sequence call fails loudly instead of reading it
  launch_prep_images(im1, im2, B, H, W, img4_.p, st, B);
  if (precision >= 1) launch_init_coords_sf(flow_init, B, H8, W8, coords1_.p, flow4_.p, x_.p, XLD, 254, st);
  else launch_init_coords(flow_init, B, H8, W8, coords1_.p, flow4_.p, x_.p + 254, XLD, st);
  launch_body(B, iters, st);
  launch_upsample(mask_.p, flow4_.p, B, H8, W8, flow_low, flow_up, st);
}

void GmaNet::forward_predictions(const float* im1, const float* im2, int B, int iters, const float* flow_init, float* preds,
                                 hipStream_t st) {
  ATDN_CHECK(ready_, "weights not finalized");
  ATDN_CHECK(B >= 1 && B <= maxB, "batch exceeds max_batch of this handle");
  ATDN_CHECK(iters >= 1 && iters <= 64, "iters out of range");
  ATDN_CHECK(im1 && im2 && preds, "null tensor");
  seq_ = 0;
  last_frame_ = 0;
  last_B_ = B;
  launch_prep_images(im1, im2, B, H, W, img4_.p, st, B);
  if (precision >= 1) launch_init_coords_sf(flow_init, B, H8, W8, coords1_.p, flow4_.p, x_.p, XLD, 254, st);
  else launch_init_coords(flow_init, B, H8, W8, coords1_.p, flow4_.p, x_.p + 254, XLD, st);
  struct Guard {   // the body reads the destination from the handle: never leave it set behind an exception
    GmaNet* n; ~Guard() { n->preds_out_ = nullptr; n->preds_stride_ = 0; }
  } guard{this};
  preds_out_ = preds;
  preds_stride_ = (long)B * 2 * H * W;
  if (precision >= 1) { FastGuard fg(precision == 2); run_body_sf(B, iters, st); } else run_body(B, iters, st);
}

BrickPyramid GmaNet::brick_pyramid() const {
  BrickPyramid bp;
  for (int l = 0; l < 4; ++l) {
    bp.base[l] = pyr_[l].p; bp.H[l] = pyrH_[l]; bp.W[l] = pyrW_[l];
    bp.BW[l] = brickBW_[l]; bp.BH[l] = brickBH_[l]; bp.NB[l] = brickNB_[l];
  }
  bp.N = N; bp.NPB = brick_pixel_blocks(N);
  return bp;
}

long GmaNet::debug_read(const char* name, float* host, long capacity, hipStream_t st) {
  const std::string k(name);
  if (!classic_ && k.size() == 4 && k.compare(0, 3, "pyr") == 0 && k[3] >= '0' && k[3] <= '3') {
    // bricked level -> the reference's row-major [pixel][H_l * W_l]
    const int l = k[3] - '0';
    const long rows = (long)maxB * N, total = rows * pyrH_[l] * pyrW_[l];
    if (scratch_.n < total) { scratch_.release(); scratch_.alloc(total); }
    launch_unbrick(pyr_[l].p, brickNB_[l], N, pyrH_[l], pyrW_[l], rows, scratch_.p, st);
    ATDN_HIP(hipStreamSynchronize(st));
    const long n = std::min(capacity, total);
    ATDN_HIP(hipMemcpy(host, scratch_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
  }
  if (!classic_ && k == "corrfeat") {
    // the fused kernel keeps the samples on chip: recompute them at the coordinates the last lookup used — for the pairs of the
    // last forward only: the pyramid and the coordinates of the handle's other pairs are whatever an earlier call (or the
    // allocator) left there, and sampling that raised the format's saturation alarm behind a parity test (round 5)
    launch_lookup_bricks(brick_pyramid(), coords_used_.p, (long)(last_B_ > 0 ? last_B_ : maxB) * N, corrfeat_.p, st);
  }
  if (k == "sf_clamped") {   // values the split-f16 format had to clamp (|x| > 65504 or NaN) since the last read
    ATDN_CHECK(capacity >= 1, "sf_clamped needs room for one float");
    host[0] = (float)sf_counter_read_reset(st);
    return 1;
  }
  const DeviceBuf* b = nullptr;
  if (k == "fmap") b = &fmap_; else if (k == "pyr0") b = &pyr_[0]; else if (k == "pyr1") b = &pyr_[1];
  else if (k == "pyr2") b = &pyr_[2]; else if (k == "pyr3") b = &pyr_[3]; else if (k == "net") b = &h_[0];
  else if (k == "x") b = &x_; else if (k == "attn") b = &attn_; else if (k == "corrfeat") b = &corrfeat_;
  else if (k == "mask") b = &mask_; else if (k == "coords1") b = &coords1_; else if (k == "flow4") b = &flow4_;
  else if (k == "qk") b = &qk_; else if (k == "img4") b = &img4_;
  else if (k == "cor1") b = &cor1_;   // relu(convc1(lookup)) of the last iteration: the fused kernel's product phase
  if (!b) return -1;
  long n = std::min(capacity, b->n);
  if (k == "attn" && !classic_) {   // fragment-major exp(s - max) + row sums -> normalised fp32 rows [maxB][N][ldN]
    const long rows = (long)maxB * N * ldN;
    if (scratch_.n < rows) { scratch_.release(); scratch_.alloc(rows); }
    ATDN_HIP(hipMemsetAsync(scratch_.p, 0, (size_t)rows * sizeof(float), st));
    launch_attn_decode(attn_.p, rinv_.p, attn_geom(maxB, N, ldN), scratch_.p, st);
    ATDN_HIP(hipStreamSynchronize(st));
    n = std::min(capacity, rows);
    ATDN_HIP(hipMemcpy(host, scratch_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
  }
  const bool is_sf = precision >= 1 && (k == "fmap" || k == "net" || k == "x" || k == "attn" || k == "corrfeat" || k == "qk" || k == "cor1");
  const float* src = b->p;
  if (is_sf) {  // decode the split-f16 tensor into a scratch fp32 copy first
    if (scratch_.n < b->n) { scratch_.release(); scratch_.alloc(b->n); }
    launch_from_sf(b->p, scratch_.p, b->n / 32, 32, st);
    src = scratch_.p;
  }
  ATDN_HIP(hipStreamSynchronize(st));
  ATDN_HIP(hipMemcpy(host, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return n;
}

}  // namespace atdn
