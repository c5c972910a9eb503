// extern "C" boundary of libatdn_hip (see include/atdn_hip.h).
#include "../../include/atdn_hip.h"

#include <cmath>

#include "clvo.h"
#include "gma.h"
#include "vae.h"
#include "clvo_train.h"
#include "frontend.h"
#include "conv_sf.h"
#include "epilogues_sf.h"

namespace atdn {
extern template TileChoice conv_dispatch<MODE_TAP, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiBias<ACT_RELU>>(const ConvShape&, EpiBias<ACT_RELU>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_ROW, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_ROW, EpiBias<ACT_RELU>>(const ConvShape&, EpiBias<ACT_RELU>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiScale>(const ConvShape&, EpiScale, hipStream_t);

extern template TileChoice conv_sf_dispatch<EpiBias<ACT_NONE>>(const ConvShape&, float, EpiBias<ACT_NONE>, hipStream_t);
extern template TileChoice conv_sf_dispatch<SfBias<ACT_NONE>>(const ConvShape&, float, SfBias<ACT_NONE>, hipStream_t);

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
}  // namespace atdn

using namespace atdn;

struct atdn_gma { GmaNet net; atdn_gma(int H, int W, int B, int prec) : net(H, W, B, prec) {} };
struct atdn_clvo { ClvoNet net; atdn_clvo(int H, int W, int B) : net(H, W, B) {} };
struct atdn_vae { VaeEncoder net; atdn_vae(int H, int W, int B) : net(H, W, B) {} };
struct atdn_ingest { FrameIngest in; atdn_ingest(int a, int b, int c, int d, int n, int aa) : in(a, b, c, d, n, aa) {} };
struct atdn_clvo_trainer { ClvoTrainer net; atdn_clvo_trainer(int H, int W, int B, int T) : net(H, W, B, T) {} };

#define ATDN_API_BEGIN try {
#define ATDN_API_END                                      \
  return 0;                                               \
  } catch (const std::exception& e) {                     \
    set_last_error(e.what());                             \
    return 1;                                             \
  } catch (...) {                                         \
    set_last_error("unknown error");                      \
    return 1;                                             \
  }

// host pose algebra helpers
template <class T>
static void euler_yxz(const T* r, T* R) {  // transforms.py:79-81
  const T c1 = std::cos(r[0]), c2 = std::cos(r[1]), c3 = std::cos(r[2]);
  const T s1 = std::sin(r[0]), s2 = std::sin(r[1]), s3 = std::sin(r[2]);
  R[0] = c1 * c3 + s1 * s2 * s3; R[1] = c3 * s1 * s2 - c1 * s3; R[2] = c2 * s1;
  R[3] = c2 * s3;                R[4] = c2 * c3;                R[5] = -s2;
  R[6] = c1 * s2 * s3 - c3 * s1; R[7] = c1 * c3 * s2 + s1 * s3; R[8] = c1 * c2;
}
template <class T>
static void make_transform(const T* rot, const T* tr, T* m) {
  T R[9];
  euler_yxz(rot, R);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[i * 4 + j] = R[i * 3 + j]; m[i * 4 + 3] = tr[i]; }
  m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1;
}
template <class T>
static void matmul4(const T* a, const T* b, T* c) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      T s = 0;
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      c[i * 4 + j] = s;
    }
}

extern "C" {

int atdn_version(void) { return 100; }
const char* atdn_last_error(void) { return g_last_error.c_str(); }

int atdn_gma_create(atdn_gma** out, int H, int W, int max_batch, int precision) {
  ATDN_API_BEGIN
  ATDN_CHECK(out, "null out pointer");
  *out = new atdn_gma(H, W, max_batch, precision);
  ATDN_API_END
}
int atdn_gma_set_low_latency(atdn_gma* h, int on) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.set_low_latency(on != 0);
  ATDN_API_END
}
int atdn_gma_load(atdn_gma* h, const char* key, const float* data, const int64_t* shape, int rank) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && key && data && rank >= 0 && rank <= 4, "bad state-dict entry");
  h->net.state().put(key, data, shape, rank);
  ATDN_API_END
}
int atdn_gma_finalize(atdn_gma* h) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.finalize();
  ATDN_API_END
}
int atdn_gma_forward(atdn_gma* h, const float* im1, const float* im2, int B, int iters, const float* flow_init,
                     float* flow_low, float* flow_up, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.forward(im1, im2, B, iters, flow_init, flow_low, flow_up, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_gma_forward_predictions(atdn_gma* h, const float* im1, const float* im2, int B, int iters, const float* flow_init,
                                 float* flow_predictions, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.forward_predictions(im1, im2, B, iters, flow_init, flow_predictions, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_gma_forward_sequence(atdn_gma* h, const float* frames, int B, int iters, const float* flow_init,
                              float* flow_low, float* flow_up, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.forward_sequence(frames, B, iters, flow_init, flow_low, flow_up, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_gma_forward_sequence_continued(atdn_gma* h, const float* frames, int B, int iters, const float* flow_init,
                                        float* flow_low, float* flow_up, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.forward_sequence(frames, B, iters, flow_init, flow_low, flow_up, (hipStream_t)stream, true);
  ATDN_API_END
}
long atdn_gma_debug_read(atdn_gma* h, const char* name, float* host, long capacity, void* stream) {
  try {
    if (!h || !name || !host) throw Error("null argument");
    const long n = h->net.debug_read(name, host, capacity, (hipStream_t)stream);
    if (n < 0) throw Error(std::string("unknown tensor name: ") + name);
    return n;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}
int atdn_gma_profile(atdn_gma* h, int B, int iters, int reps, float* ms_out, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && ms_out, "null argument");
  static_assert(GmaNet::ST_COUNT == ATDN_GMA_STAGES, "stage table out of sync with the header");
  h->net.profile(B, iters, reps, ms_out, (hipStream_t)stream, 0);
  ATDN_API_END
}
int atdn_gma_profile_mode(atdn_gma* h, int B, int iters, int reps, int mode, float* ms_out, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && ms_out, "null argument");
  ATDN_CHECK(mode >= 0 && mode <= 2, "mode must be 0 (pair), 1 (sequence) or 2 (continued sequence)");
  h->net.profile(B, iters, reps, ms_out, (hipStream_t)stream, mode);
  ATDN_API_END
}
size_t atdn_gma_workspace_bytes(atdn_gma* h) { return h ? h->net.workspace_bytes() : 0; }
void atdn_gma_destroy(atdn_gma* h) { delete h; }

int atdn_clvo_create(atdn_clvo** out, int H, int W, int max_batch) {
  ATDN_API_BEGIN
  ATDN_CHECK(out, "null out pointer");
  *out = new atdn_clvo(H, W, max_batch);
  ATDN_API_END
}
int atdn_clvo_load(atdn_clvo* h, const char* key, const float* data, const int64_t* shape, int rank) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && key && data && rank >= 0 && rank <= 4, "bad state-dict entry");
  h->net.state().put(key, data, shape, rank);
  ATDN_API_END
}
int atdn_clvo_finalize(atdn_clvo* h) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.finalize();
  ATDN_API_END
}
int atdn_clvo_encode(atdn_clvo* h, const float* flow, int B, float* feat, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && flow && feat, "null argument");
  h->net.encode(flow, B, feat, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_clvo_step(atdn_clvo* h, const float* feat, int T, int Bs, float* state, float* rot, float* tr, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && feat && state && rot && tr, "null argument");
  h->net.step(feat, T, Bs, state, rot, tr, (hipStream_t)stream);
  ATDN_API_END
}
void atdn_clvo_destroy(atdn_clvo* h) { delete h; }

int atdn_vae_create(atdn_vae** out, int H, int W, int max_batch) {
  ATDN_API_BEGIN
  ATDN_CHECK(out, "null out pointer");
  *out = new atdn_vae(H, W, max_batch);
  ATDN_API_END
}
int atdn_vae_load(atdn_vae* h, const char* key, const float* data, const int64_t* shape, int rank) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && key && data && rank >= 0 && rank <= 4, "bad state-dict entry");
  h->net.state().put(key, data, shape, rank);
  ATDN_API_END
}
int atdn_vae_finalize(atdn_vae* h) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.finalize();
  ATDN_API_END
}
int atdn_vae_embedding_shape(const atdn_vae* h, int* out_h, int* out_w) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && out_h && out_w, "null argument");
  *out_h = h->net.out_h();
  *out_w = h->net.out_w();
  ATDN_API_END
}
int atdn_vae_encode(atdn_vae* h, const float* images, int B, float* mu, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && images && mu, "null argument");
  h->net.encode(images, B, mu, (hipStream_t)stream);
  ATDN_API_END
}
void atdn_vae_destroy(atdn_vae* h) { delete h; }

int atdn_clvo_trainer_create(atdn_clvo_trainer** out, int H, int W, int batch, int sequence_length) {
  ATDN_API_BEGIN
  ATDN_CHECK(out, "null out pointer");
  *out = new atdn_clvo_trainer(H, W, batch, sequence_length);
  ATDN_API_END
}
int atdn_clvo_trainer_load(atdn_clvo_trainer* h, const char* key, const float* data, const int64_t* shape, int rank) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && key && data && rank >= 0 && rank <= 4, "bad state-dict entry");
  h->net.state().put(key, data, shape, rank);
  ATDN_API_END
}
int atdn_clvo_trainer_finalize(atdn_clvo_trainer* h) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.finalize();
  ATDN_API_END
}
int atdn_clvo_trainer_forward_backward(atdn_clvo_trainer* h, const float* flows, const float* true_rot, const float* true_tr,
                                       float* pred_rot, float* pred_tr, float* loss_out, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && flows && true_rot && true_tr && loss_out, "null argument");
  *loss_out = h->net.forward_backward(flows, true_rot, true_tr, pred_rot, pred_tr, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_clvo_trainer_gradients(atdn_clvo_trainer* h, float** device_ptr, long* count) {
  ATDN_API_BEGIN
  ATDN_CHECK(h && device_ptr && count, "null argument");
  *device_ptr = h->net.grad_buffer();
  *count = h->net.grad_count();
  ATDN_API_END
}
int atdn_clvo_trainer_adamw_step(atdn_clvo_trainer* h, float lr, float weight_decay, float eps, int step, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->net.adamw_step(lr, weight_decay, eps, step, (hipStream_t)stream);
  ATDN_API_END
}
long atdn_clvo_trainer_read(atdn_clvo_trainer* h, const char* key, int kind, float* host, long capacity, void* stream) {
  try {
    ATDN_CHECK(h && key && host, "null argument");
    return h->net.read(key, kind, host, capacity, (hipStream_t)stream);
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}
void atdn_clvo_trainer_destroy(atdn_clvo_trainer* h) { delete h; }

// ------------------------------------------------------------------ pose algebra (host)
int atdn_pose_transform_f32(const float* rot, const float* tr, float* mat16) {
  ATDN_API_BEGIN
  ATDN_CHECK(rot && tr && mat16, "null argument");
  make_transform(rot, tr, mat16);
  ATDN_API_END
}
int atdn_pose_rel2abs(const float* rot, const float* tr, int T, double* poses) {
  ATDN_API_BEGIN
  ATDN_CHECK(rot && tr && poses && T >= 0, "bad argument");
  for (int i = 0; i < 16; ++i) poses[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int t = 0; t < T; ++t) {
    const double r[3] = {rot[t * 3], rot[t * 3 + 1], rot[t * 3 + 2]};
    const double x[3] = {tr[t * 3], tr[t * 3 + 1], tr[t * 3 + 2]};
    double m[16];
    make_transform(r, x, m);
    matmul4(poses + (long)t * 16, m, poses + (long)(t + 1) * 16);
  }
  ATDN_API_END
}
int atdn_pose_accumulate_f32(float* pose16, const float* rot, const float* tr) {
  ATDN_API_BEGIN
  ATDN_CHECK(pose16 && rot && tr, "null argument");
  float m[16], o[16];
  make_transform(rot, tr, m);
  matmul4(pose16, m, o);
  for (int i = 0; i < 16; ++i) pose16[i] = o[i];
  ATDN_API_END
}

// ------------------------------------------------------------------ individual kernels
int atdn_corr_lookup(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3, int B, int H8,
                     int W8, const float* coords, float* out, int ldo, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(pyr0 && pyr1 && pyr2 && pyr3 && coords && out && B >= 1, "null argument");
  ATDN_CHECK((H8 >> 3) >= 2 && (W8 >> 3) >= 2, "map too small for four pyramid levels");
  PyramidLevels pl;
  const float* b[4] = {pyr0, pyr1, pyr2, pyr3};
  for (int l = 0; l < 4; ++l) { pl.base[l] = b[l]; pl.H[l] = H8 >> l; pl.W[l] = W8 >> l; }
  launch_lookup(pl, coords, (long)B * H8 * W8, out, ldo, (hipStream_t)stream);
  ATDN_API_END
}

int atdn_corr_pyramid(const float* fmap1, const float* fmap2, int B, int H8, int W8, int C, float* pyr0, float* pyr1,
                      float* pyr2, float* pyr3, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(fmap1 && fmap2 && pyr0 && pyr1 && pyr2 && pyr3 && B >= 1 && C % 32 == 0, "bad argument");
  const int N = H8 * W8;
  ConvShape c;
  c.src0 = fmap1; c.ld0 = C; c.sb0 = (long)N * C; c.C0 = C; c.H = 1; c.W = N;
  c.w = fmap2; c.wb = (long)N * C; c.ldw = C; c.N = N; c.nimg = B;
  conv_dispatch<MODE_TAP>(c, EpiScale{1.0f / sqrtf((float)C), pyr0, (long)N * N, N}, (hipStream_t)stream);
  float* p[4] = {pyr0, pyr1, pyr2, pyr3};
  for (int l = 1; l < 4; ++l) launch_avgpool(p[l - 1], H8 >> (l - 1), W8 >> (l - 1), p[l], (long)B * N, (hipStream_t)stream);
  ATDN_API_END
}

// The PRODUCT kernels of the default (split-f16) path on caller-supplied features — corr_bricks_kernel for the four levels
// (levels 1-3 from 2x2-pooled target features), the brick-major pyramid, lookup_conv_kernel in both instantiations — so that the
// reference's own CorrBlock probe set reaches the kernels the benchmark times. Scratch is allocated per call:
// a unit-test entry, not a hot path.
int atdn_corr_lookup_bricks(const float* fmap1, const float* fmap2, int B, int H8, int W8, int C, const float* coords,
                            float* pyr0, float* pyr1, float* pyr2, float* pyr3, float* samples,
                            const float* convc1_weight_host, const float* convc1_bias_host, float* cor1, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(fmap1 && fmap2 && B >= 1 && H8 >= 1 && W8 >= 1, "bad argument");
  ATDN_CHECK(C == 256, "the correlation kernel keeps a source strip's whole K = 256 in registers: C must be 256");
  ATDN_CHECK((H8 >> 3) >= 2 && (W8 >> 3) >= 2, "map too small for four pyramid levels");
  ATDN_CHECK(!(samples || cor1) || coords, "a lookup needs coordinates");
  ATDN_CHECK(!cor1 || (convc1_weight_host && convc1_bias_host), "cor1 needs convc1's weight [256][324] and bias [256]");
  hipStream_t st = (hipStream_t)stream;
  const int N = H8 * W8;
  const long n8 = (long)B * N;
  BrickPyramid bp;
  bp.N = N; bp.NPB = brick_pixel_blocks(N);
  DeviceBuf f1, f2, plain[3], fbrick[4], pyr[4], scratch, outsf;
  DeviceBuf* all[] = {&f1, &f2, &plain[0], &plain[1], &plain[2], &fbrick[0], &fbrick[1], &fbrick[2], &fbrick[3],
                      &pyr[0], &pyr[1], &pyr[2], &pyr[3], &scratch, &outsf};
  WeightArena A;
  try {
    sf_counter_attach();
    f1.alloc(n8 * 256); f2.alloc(n8 * 256);
    launch_to_sf(fmap1, f1.p, n8, 256, st);
    launch_to_sf(fmap2, f2.p, n8, 256, st);
    float* rowmajor[4] = {pyr0, pyr1, pyr2, pyr3};
    int pH[4], pW[4];
    for (int l = 0; l < 4; ++l) {   // the same sequence as GmaNet::run_body_sf
      pH[l] = H8 >> l; pW[l] = W8 >> l;
      bp.H[l] = pH[l]; bp.W[l] = pW[l]; bp.BW[l] = cdiv(pW[l], 8); bp.BH[l] = cdiv(pH[l], 4); bp.NB[l] = bp.BW[l] * bp.BH[l] * 32;
      const float* src = f2.p;
      long src_sb = (long)N * 256;
      if (l > 0) {
        const float* prev = l == 1 ? f2.p : plain[l - 2].p;
        const long prev_sb = l == 1 ? (long)N * 256 : (long)pH[l - 1] * pW[l - 1] * 256;
        src_sb = (long)pH[l] * pW[l] * 256;
        plain[l - 1].alloc((long)B * src_sb);
        launch_pool_features_sf(prev, B, pH[l - 1], pW[l - 1], 256, prev_sb, plain[l - 1].p, src_sb, st);
        src = plain[l - 1].p;
      }
      fbrick[l].alloc((long)B * bp.NB[l] * 256);
      launch_brick_rows(src, src_sb, B, pH[l], pW[l], 256, fbrick[l].p, (long)bp.NB[l] * 256, st);
      pyr[l].alloc((long)B * bp.NPB * kBrickPixelBlock * bp.NB[l]);
      launch_corr_bricks(f1.p, (long)N * 256, fbrick[l].p, (long)bp.NB[l] * 256, B, N, bp.NB[l], 1.0f / sqrtf(256.0f), pyr[l].p, false, st);
      bp.base[l] = pyr[l].p;
      if (rowmajor[l]) launch_unbrick(pyr[l].p, bp.NB[l], N, pH[l], pW[l], n8, rowmajor[l], st);
    }
    if (samples) {   // lookup_conv_kernel<FUSED = false>: the sampling code of the fused kernel, samples to memory
      outsf.alloc(n8 * 352);
      launch_lookup_bricks(bp, coords, n8, outsf.p, st);
      scratch.alloc(n8 * 352);
      launch_from_sf(outsf.p, scratch.p, n8, 352, st);
      ATDN_HIP(hipMemcpy2DAsync(samples, 324 * sizeof(float), scratch.p, 352 * sizeof(float), 324 * sizeof(float), (size_t)n8,
                                hipMemcpyDeviceToDevice, st));
    }
    if (cor1) {      // lookup_conv_kernel<FUSED = true>: what GmaNet::iteration_sf launches
      StateDict sd;
      const int64_t ws[4] = {256, 324, 1, 1}, bs[1] = {256};
      sd.put("c.weight", convc1_weight_host, ws, 4);
      sd.put("c.bias", convc1_bias_host, bs, 1);
      PackedConv L = pack_conv_sf(A, sd, {"c"});
      pack_fragment_major16(A, L);
      A.upload();
      resolve(A, L);
      DeviceBuf csf;
      csf.alloc(n8 * 256);
      try {
        launch_lookup_conv(bp, coords, n8, nullptr, L.wf16, L.wscale, L.b, csf.p, false, st);
        launch_from_sf(csf.p, cor1, n8, 256, st);
        ATDN_HIP(hipStreamSynchronize(st));
      } catch (...) { csf.release(); throw; }
      csf.release();
    }
    ATDN_HIP(hipStreamSynchronize(st));
  } catch (...) {
    (void)hipDeviceSynchronize();
    for (auto* b : all) b->release();
    A.release();
    throw;
  }
  for (auto* b : all) b->release();
  A.release();
  ATDN_API_END
}

int atdn_conv2d_nhwc(const float* src, int nimg, int H, int W, int Cin, const float* weight_host,
                     const float* bias_host, int Cout, int KH, int KW, int stride, int padH, int padW, int relu,
                     float* dst, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(src && weight_host && dst && nimg >= 1, "null argument");
  const bool row = (Cin == 4 || Cin == 16);
  ATDN_CHECK(row || Cin % 32 == 0, "Cin must be 4, 16 or a multiple of 32");
  StateDict sd;
  const int64_t ws[4] = {Cout, Cin, KH, KW};
  sd.put("c.weight", weight_host, ws, 4);
  std::vector<float> zb(Cout, 0.f);
  const int64_t bs[1] = {Cout};
  sd.put("c.bias", bias_host ? bias_host : zb.data(), bs, 1);
  WeightArena A;
  PackedConv L = pack_conv(A, sd, {"c"}, row ? MODE_ROW : MODE_TAP, Cin);
  A.upload();
  resolve(A, L);
  ConvShape s;
  s.src0 = src; s.ld0 = Cin; s.sb0 = (long)H * W * Cin; s.C0 = L.C; s.H = H; s.W = W;
  s.KH = KH; s.KW = KW; s.stride = stride; s.padH = padH; s.padW = padW;
  s.w = L.w; s.ldw = L.ldw; s.N = Cout; s.nimg = nimg;
  const int Ho = conv_out(H, KH, stride, padH), Wo = conv_out(W, KW, stride, padW);
  hipStream_t st = (hipStream_t)stream;
  try {
    if (row) {
      if (relu) conv_dispatch<MODE_ROW>(s, EpiBias<ACT_RELU>{L.b, dst, (long)Ho * Wo * Cout, Cout, 1.f}, st);
      else conv_dispatch<MODE_ROW>(s, EpiBias<ACT_NONE>{L.b, dst, (long)Ho * Wo * Cout, Cout, 1.f}, st);
    } else {
      if (relu) conv_dispatch<MODE_TAP>(s, EpiBias<ACT_RELU>{L.b, dst, (long)Ho * Wo * Cout, Cout, 1.f}, st);
      else conv_dispatch<MODE_TAP>(s, EpiBias<ACT_NONE>{L.b, dst, (long)Ho * Wo * Cout, Cout, 1.f}, st);
    }
    ATDN_HIP(hipStreamSynchronize(st));
  } catch (...) {
    A.release();
    throw;
  }
  A.release();
  ATDN_API_END
}

int atdn_resize_frames(const float* src, int planes, int Hin, int Win, int Hout, int Wout, float* dst, void* stream) {
  return atdn_resize_frames_mode(src, planes, Hin, Win, Hout, Wout, ATDN_RESIZE_ANTIALIAS, dst, stream);
}
int atdn_resize_frames_mode(const float* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst,
                            void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(src && dst && planes >= 1 && Hin >= 1 && Win >= 1 && Hout >= 1 && Wout >= 1, "bad argument");
  ATDN_CHECK(antialias == 0 || antialias == 1, "antialias must be ATDN_RESIZE_BILINEAR or ATDN_RESIZE_ANTIALIAS");
  launch_resize<float>(src, planes, Hin, Win, Hout, Wout, antialias, dst, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_resize_frames_u8(const uint8_t* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst,
                          void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(src && dst && planes >= 1 && Hin >= 1 && Win >= 1 && Hout >= 1 && Wout >= 1, "bad argument");
  ATDN_CHECK(antialias == 0 || antialias == 1, "antialias must be ATDN_RESIZE_BILINEAR or ATDN_RESIZE_ANTIALIAS");
  launch_resize<unsigned char>(src, planes, Hin, Win, Hout, Wout, antialias, dst, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_pad_frames(const float* src, int planes, int H, int W, int left, int right, int top, int bottom, float* dst,
                    void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(src && dst && planes >= 1 && H >= 1 && W >= 1 && left >= 0 && right >= 0 && top >= 0 && bottom >= 0, "bad argument");
  launch_pad_replicate(src, planes, H, W, left, right, top, bottom, dst, (hipStream_t)stream);
  ATDN_API_END
}
int atdn_ingest_create(atdn_ingest** out, int Hin, int Win, int Hout, int Wout, int max_frames, int antialias) {
  ATDN_API_BEGIN
  ATDN_CHECK(out, "null out pointer");
  ATDN_CHECK(antialias == 0 || antialias == 1, "antialias must be ATDN_RESIZE_BILINEAR or ATDN_RESIZE_ANTIALIAS");
  *out = new atdn_ingest(Hin, Win, Hout, Wout, max_frames, antialias);
  ATDN_API_END
}
int atdn_ingest_frames_u8(atdn_ingest* h, const uint8_t* host_frames, int n_frames, float* dst, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(h, "null handle");
  h->in.ingest(host_frames, n_frames, dst, (hipStream_t)stream);
  ATDN_API_END
}
void atdn_ingest_destroy(atdn_ingest* h) { delete h; }

int atdn_conv2d_nhwc_sf_epi(const float* src, int nimg, int H, int W, int Cin, const float* weight_host,
                            const float* bias_host, int Cout, int KH, int KW, int stride, int padH, int padW, int sf_store,
                            float* dst, void* stream) {
  ATDN_API_BEGIN
  ATDN_CHECK(src && weight_host && dst && nimg >= 1 && Cin % 32 == 0, "bad argument (Cin must be a multiple of 32)");
  ATDN_CHECK(!sf_store || Cout % 32 == 0, "the split-f16 store needs Cout % 32 == 0");
  StateDict sd;
  const int64_t ws[4] = {Cout, Cin, KH, KW};
  sd.put("c.weight", weight_host, ws, 4);
  std::vector<float> zb(Cout, 0.f);
  const int64_t bs[1] = {Cout};
  sd.put("c.bias", bias_host ? bias_host : zb.data(), bs, 1);
  WeightArena A;
  PackedConv L = pack_conv_sf(A, sd, {"c"});
  A.upload();
  resolve(A, L);
  hipStream_t st = (hipStream_t)stream;
  float* tmp = nullptr;
  const long rows = (long)nimg * H * W;
  ATDN_HIP(hipMalloc(&tmp, (size_t)rows * Cin * sizeof(float)));
  try {
    launch_to_sf(src, tmp, rows, Cin, st);
    ConvShape s;
    s.src0 = tmp; s.ld0 = Cin; s.sb0 = (long)H * W * Cin; s.C0 = L.C; s.H = H; s.W = W;
    s.KH = KH; s.KW = KW; s.stride = stride; s.padH = padH; s.padW = padW;
    s.w = L.w; s.ldw = L.ldw; s.N = Cout; s.nimg = nimg;
    const int Ho = conv_out(H, KH, stride, padH), Wo = conv_out(W, KW, stride, padW);
    s.wfrag16 = L.wf16;
    // sf_store: write split-f16 through the SfBias epilogue (the channel-vector sf store of the product's layers), then
    // unpack to fp32; otherwise fp32 output through EpiBias
    if (sf_store) {
      float* osf = nullptr;
      const long orows = (long)nimg * Ho * Wo;
      ATDN_HIP(hipMalloc(&osf, (size_t)orows * Cout * sizeof(float)));
      try {
        conv_sf_dispatch(s, L.wscale, SfBias<ACT_NONE>{L.b, osf, (long)Ho * Wo * Cout, Cout}, st);
        launch_from_sf(osf, dst, orows, Cout, st);
        ATDN_HIP(hipStreamSynchronize(st));
      } catch (...) {
        (void)hipFree(osf);
        throw;
      }
      (void)hipFree(osf);
    } else {
      conv_sf_dispatch(s, L.wscale, EpiBias<ACT_NONE>{L.b, dst, (long)Ho * Wo * Cout, Cout, 1.f}, st);
      ATDN_HIP(hipStreamSynchronize(st));
    }
  } catch (...) {
    (void)hipFree(tmp);
    A.release();
    throw;
  }
  (void)hipFree(tmp);
  A.release();
  ATDN_API_END
}

int atdn_conv2d_nhwc_sf(const float* src, int nimg, int H, int W, int Cin, const float* weight_host,
                        const float* bias_host, int Cout, int KH, int KW, int stride, int padH, int padW, float* dst,
                        void* stream) {
  return atdn_conv2d_nhwc_sf_epi(src, nimg, H, W, Cin, weight_host, bias_host, Cout, KH, KW, stride, padH, padW, 0, dst, stream);
}

}  // extern "C"
