// GMA optical-flow forward (RAFTGMA.forward, test_mode=True) as one hipGraph of hand-written kernels.
#pragma once
#include <vector>
#include <map>
#include <memory>

#include "attention.h"
#include "conv_dispatch.h"
#include "kernels.h"
#include "weights.h"

namespace atdn {

struct EncoderWeights {
  PackedConv stem;
  struct Block { PackedConv c1, c2, ds; bool has_ds = false; } blk[6];
  PackedConv head;
};

class DeviceBuf {
 public:
  float* p = nullptr;
  long n = 0;
  void alloc(long count) {
    n = count;
    ATDN_HIP(hipMalloc(&p, (size_t)count * sizeof(float)));
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; } }
};

class GmaNet {
 public:
  // precision: 0 = exact-fp32 MFMA everywhere; 1 = split-f16 (3 x f16 MFMA, fp32-grade) for every TAP-mode GEMM
  GmaNet(int H, int W, int max_batch, int precision);
  ~GmaNet();
  StateDict& state() { return sd_; }
  void finalize();  // pack + upload weights, allocate the workspace
  // im1/im2 NCHW [B,3,H,W] 0..255; flow_init NCHW [B,2,H/8,W/8] or null; outputs NCHW. All device pointers.
  void forward(const float* im1, const float* im2, int B, int iters, const float* flow_init, float* flow_low,
               float* flow_up, hipStream_t st);
  // B consecutive pairs of one clip: frames NCHW [B+1,3,H,W]; pair b = (frame b, frame b+1). The feature network
  // runs once per FRAME (B+1 passes instead of 2B).
  // flow_predictions of network.py:106-129 (test_mode=False): the convex upsampling of EVERY iteration's flow with that
  // iteration's mask, preds [iters][B][2][H][W]. The mask head runs per iteration here (in test mode only the last one's
  // reaches the output); launched eagerly, not as a graph: the destination moves with the iteration.
  void forward_predictions(const float* im1, const float* im2, int B, int iters, const float* flow_init, float* preds,
                           hipStream_t st);
  void forward_sequence(const float* frames, int B, int iters, const float* flow_init, float* flow_low, float* flow_up,
                        hipStream_t st, bool continued = false);
  // copy an internal tensor to host (parity tests); returns element count, or -1 for an unknown name
  long debug_read(const char* name, float* host, long capacity, hipStream_t st);
  size_t workspace_bytes() const { return ws_bytes_; }
  // Eager (un-graphed) run on `st` with hipEvents at stage boundaries; ms[] receives the time of each Stage
  // summed over `reps` forwards. Inputs are whatever the last forward() left in the workspace.
  enum Stage { ST_FNET = 0, ST_CORR, ST_POOL, ST_CNET, ST_ATTN, ST_LOOKUP, ST_MOTION, ST_AGG, ST_GRU_ZR, ST_GRU_Q, ST_FLOWHEAD,
               ST_MASK, ST_GRU_CTX, ST_ATTN_LOGITS, ST_AGG_VT, ST_CONVC1, ST_GRU_ZR_V, ST_GRU_Q_V, ST_COUNT };
  void profile(int B, int iters, int reps, float* ms, hipStream_t st, int mode = 0);

  int H, W, H8, W8, N, ldN, maxB, precision;
  int dev_ = 0;   // device the handle lives on
  // feature network, split-f16 pipeline (precision 1): conv2 of every residual block normalises conv1's raw output in its own
  // patch loader (conv_sf6.h NORM); the f16 fast mode keeps the separate normalisation pass
  bool norm_on_load_ = false;
  // precision 0 (exact-fp32 MFMA everywhere) keeps the classical data flow: row-major correlation pyramid pooled from the
  // level-0 volume, separate lookup and convc1, fp32 logits + softmax pass + attention x V on the GEMM engine. The split-f16
  // modes use the bricked pyramid (every level a GEMM against pooled features in brick order, 4 x 8 cells = one 128-byte
  // line), the lookup fused with convc1 (lookup_fused.hip) and the fused attention kernels (attention.hip).
  bool classic_ = false;
  // Low-latency form for the per-frame callers (neural_slam.py:202 calls the flow network with ONE pair per frame): launches
  // that would leave most of the chip idle at small B are cut finer — today attention x V, along its key axis (attention.h).
  // Another summation order than the default path, so it is opt-in (set before finalize()); the default path keeps clip mode,
  // continued clips and pair mode bit-identical.
  bool low_latency_ = false;
  void set_low_latency(bool on);
  DeviceBuf attn_part_;   // [8][maxB][Npad][128] fp32 partial sums of the split attention x V
  // ... and, in the captured graph of one or two pairs, independent chains run as parallel branches on a second capture
  // stream: feature network + correlation pyramid beside context network + attention + GRU context terms, and inside an
  // iteration the flow branch of the motion encoder beside the correlation branch. Same kernels, same operands (the context
  // network gets scratch maps of its own), so the same bits; at 16 pairs every launch fills the chip and branches only
  // share hardware queues with the other clip's stream (round 3 measured that: -3 % in the sequence driver).
  DeviceBuf enc2_[4];
  hipStream_t par_stream_ = nullptr;
  std::vector<hipEvent_t> par_events_;
  size_t par_next_ = 0;
  bool par_ = false;       // true only inside capture() of a low-latency handle
  bool par_ok_ = true;     // cleared when a capture with branches failed once: later captures are single chains
  void fork(hipStream_t from, hipStream_t to);   // `to` continues from where `from` stands
  DeviceBuf fbrick_[4], fplain_[3], coords_used_;   // features in brick order (levels 0-3), plain pooled features (1-3)
  int brickBW_[4], brickBH_[4], brickNB_[4];
  BrickPyramid brick_pyramid() const;
  DeviceBuf rowmax_, rinv_;

 private:
  void run_body(int B, int iters, hipStream_t st);  // everything between input prep and upsampling
  void run_encoder(const EncoderWeights& E, bool instance, int nimg, hipStream_t st, float** out_buf, int* outH,
                   int* outW);
  void iteration(int B, hipStream_t st);
  void run_body_sf(int B, int iters, hipStream_t st);
  // first_img: index of the first image of img4_ to encode (continued sequences skip frame 0 in the feature network)
  void run_encoder_sf(const EncoderWeights& E, bool instance, int nimg, hipStream_t st, float** out_buf, int first_img = 0,
                      DeviceBuf* bufs = nullptr);   // bufs: four scratch maps (default enc_)
  void iteration_sf(int B, hipStream_t st);
  void capture(int B, int iters);
  void launch_body(int B, int iters, hipStream_t st);
  int seq_ = 0;        // 0: pair mode; 1: sequence (B+1 frames, every frame through fnet once); 2: sequence continued
                       //    (frame 0 is the previous call's last frame: its features are reused, fnet sees B frames)
  int last_frame_ = -1;  // fmap_ slot of the last frame of the previous sequence call
  float* preds_out_ = nullptr;   // forward_predictions only: where iteration `it` writes its upsampled flow (+ it * preds_stride_)
  long preds_stride_ = 0;
  void mask_head_sf(int B, hipStream_t st);
  void mask_head(int B, hipStream_t st);
  int last_B_ = 0;       // pairs of the last forward (debug reads that recompute something do so for these pairs only)

  StateDict sd_;
  WeightArena arena_;
  bool ready_ = false;
  bool use_graph_ = true;
  EncoderWeights fnet_, cnet_;
  PackedConv convc1_, convc2_, convf1_, convf2_, convm_, to_v_, to_qk_;
  long convf1_sf_off_ = -1;  // convf1 weights in split-f16 fragment order for small_convs.hip (flow_conv7_sf_kernel)
  float convf1_wscale_ = 1.f;
  float fh2_mul_ = 1.f;      // power-of-two scale of the flow head's conv2 weights (split-f16 product in conv1's epilogue)
  long fh2_w32_off_ = -1;    // flow head conv2 weights as fp32 [tap*2 + output][256] for the fused flow head
  DeviceBuf fhG_;            // conv2 partial sums [maxB * N][18]
  PackedConv gru_zr_[2], gru_q_[2], fh1_, fh2_, mask0_, mask2_;
  PackedConv gru_zr_ctx_[2], gru_q_ctx_[2];  // sf mode: context-channel (inp) slices, applied once per pair
  const float* gamma_ = nullptr;
  long gamma_off_ = -1;

  // workspace
  DeviceBuf img4_, enc_[4], scratch_, pcnt_, fin_, fmap_, psum_, pm2_, mean_[3], rstd_[3];   // [2]: the feature network's stem statistics, alive until its first block's residual pass
  DeviceBuf pyr_[4], h_[2], x_, qk_, attn_, vT_, corrfeat_, cor1_, corflo_, flo1_, z_, rh_, fh_, mask_;
  DeviceBuf coords1_, flow4_, pre_zr_[2], pre_q_[2];
  int pyrH_[4], pyrW_[4];
  size_t ws_bytes_ = 0;

  struct Timer;
  Timer* timer_ = nullptr;
  void mark(int stage, hipStream_t st);

  hipStream_t cap_stream_ = nullptr;
  std::map<std::pair<int, int>, hipGraphExec_t> graphs_;  // key: (B, iters * 2 + seq)
};

}  // namespace atdn
