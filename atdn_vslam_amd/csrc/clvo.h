// CLVO pose head `ATDNVO` (atdn_vslam/odometry/network.py:122-146): stateless CNN encoder (shardable over
// frame pairs) + the sequential LSTM/MLP tail with explicit state.
#pragma once
#include "conv_dispatch.h"
#include "kernels.h"
#include "weights.h"
#include "gma.h"  // DeviceBuf
#include <set>

namespace atdn {

class ClvoNet {
 public:
  ClvoNet(int H, int W, int max_batch);
  ~ClvoNet();
  StateDict& state() { return sd_; }
  void finalize();
  // flow NCHW [B,2,H,W] -> feat [B][512]
  void encode(const float* flow, int B, float* feat, hipStream_t st);
  // feat [T][Bs][512]; state [4][Bs][512] = h1,c1,h2,c2 (in/out); rot,tr [T][Bs][3]
  void step(const float* feat, int T, int Bs, float* state, float* rot, float* tr, hipStream_t st);

  int H, W, maxB;

 private:
  // raw_*: the layer's OIHW weights and bias as the state dict holds them, for the 16x16x4 MFMA kernels (conv16)
  struct ConvBN { PackedConv conv; long sc_off = -1, sh_off = -1; const float* sc = nullptr; const float* sh = nullptr;
                  long raw_w_off = -1, raw_b_off = -1; const float* raw_w = nullptr; const float* raw_b = nullptr; };
  struct Res { ConvBN a, b; PackedConv skip; long sc_off = -1, sh_off = -1; const float* sc = nullptr; const float* sh = nullptr;
               long skip_w_off = -1, skip_b_off = -1; const float* skip_w = nullptr; const float* skip_b = nullptr; };
  ConvBN pack_convbn(const std::string& p);

  StateDict sd_;
  WeightArena arena_;
  bool ready_ = false;
  ConvBN stem_, last_;
  Res res_[4];
  long dw_w_off_ = -1, dw_b_off_ = -1;
  struct Lin { long w_off = -1, b_off = -1; const float* w = nullptr; const float* b = nullptr; };
  Lin fc_, lstm1_ih_, lstm1_hh_, lstm_lin_, lstm2_ih_, lstm2_hh_, rot_[3], tr_[3];
  Lin pack_linear(const std::string& wkey, const std::string& bkey, const std::vector<int>* perm = nullptr);

  DeviceBuf in4_, bufA_, bufB_, bufS_, flat_;
  DeviceBuf pre_, hseq_, x2seq_;  // scan scratch, grown on demand: [T*Bs][2048], [(T+1)*Bs][512], [T*Bs][512]
  DeviceBuf hseq2_;               // [(T+1)*Bs][512]: the h2 sequence of the pipelined scan
  void ensure_scan(long rows, int Bs);
  // the T + 2 dependent launches of the pipelined scan replay as ONE hipGraph per (T, Bs) once that shape has been seen
  // before (first sight and T > kMaxGraphedSteps launch eagerly: clvo.hip, "Graph policy").
  // The graph only references library-owned buffers (the caller's state is copied in and out around it).
  DeviceBuf cstate_;              // [2][Bs][512]: c1, c2 of the scan in flight
  std::map<std::pair<int, int>, hipGraphExec_t> scan_graphs_;
  std::set<std::pair<int, int>> scan_seen_;
  static constexpr int kMaxGraphedSteps = 1024;
  int dev_ = 0;                   // device the handle lives on (destructor)
  hipStream_t cap_stream_ = nullptr;
  bool scan_graph_ = true;        // ATDN_NO_GRAPH=1: eager launches
  void launch_scan_steps(int T, int Bs, hipStream_t st);
  // persistent scan (lstm_scan.hip): sequences of one batch row and at least kPersistentMinSteps steps run as ONE
  // launch. ATDN_SCAN_PERSISTENT=0 keeps the per-step kernel. A launch that gave up on a bounded spin leaves a non-zero word in
  // scan_abort_host_ (pinned; copied behind the launch): step() synchronises, sees it, restores the state it saved in
  // scan_state0_, repeats the sequence on the per-step kernel and stays there (clvo.hip). ATDN_SCAN_TEST_ABORT=1 makes the next
  // persistent launch of a handle give up at once (the test of that path).
  static constexpr int kPersistentMinSteps = 16;
  bool scan_persistent_ = true;
  DeviceBuf scan_xch_, scan_state0_;
  unsigned int* scan_abort_host_ = nullptr;
};

}  // namespace atdn
