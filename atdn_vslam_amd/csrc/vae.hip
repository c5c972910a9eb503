// MappingVAE encoder on MI355X. Reference: atdn_vslam/localization/network.py:29-45 (layers), 57-70 (forward,
// non-variational: mu = mean_lin(encoder(normalize(image)))); layers/conv.py:36-37 (Conv = BN(Mish(conv))),
// 83-90 (ResidualConv); utils/normalizations.py:4-6 (x/255, then ImageNet mean/std).
#include "vae.h"

namespace atdn {

extern template TileChoice conv_dispatch<MODE_ROW, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_ROW, EpiMishBN>(const ConvShape&, EpiMishBN, hipStream_t);
extern template TileChoice conv_dispatch<MODE_ROW, EpiMishBNSkipMishBN>(const ConvShape&, EpiMishBNSkipMishBN, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiBias<ACT_NONE>>(const ConvShape&, EpiBias<ACT_NONE>, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiMishBN>(const ConvShape&, EpiMishBN, hipStream_t);
extern template TileChoice conv_dispatch<MODE_TAP, EpiMishBNSkipMishBN>(const ConvShape&, EpiMishBNSkipMishBN, hipStream_t);

namespace {
constexpr int kCh[7] = {3, 16, 16, 32, 64, 128, 128};
inline int pix_channels(int c) { return c <= 4 ? 4 : c <= 16 ? 16 : c; }  // ROW mode pads to 4 / 16, TAP needs % 32
inline int conv_mode(int cin) { return cin >= 32 ? MODE_TAP : MODE_ROW; }

template <class Epi>
void run_conv(int mode, const ConvShape& s, const Epi& ep, hipStream_t st) {
  if (mode == MODE_TAP) conv_dispatch<MODE_TAP>(s, ep, st);
  else conv_dispatch<MODE_ROW>(s, ep, st);
}
}  // namespace

VaeEncoder::VaeEncoder(int H_, int W_, int max_batch) : H(H_), W(W_), maxB(max_batch) {
  ATDN_CHECK(max_batch >= 1 && max_batch <= 64, "max_batch out of range");
  ATDN_CHECK(H >= 64 && W >= 64 && (long)H * W <= (1L << 24), "frame size out of range");
  int h = H, w = W;
  for (int i = 0; i < 6; ++i) { h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1); }
  oh_ = h; ow_ = w;
}

VaeEncoder::~VaeEncoder() {
  for (DeviceBuf* b : {&in4_, &bufA_, &bufB_, &bufS_}) b->release();
  arena_.release();
}

VaeEncoder::ConvBN VaeEncoder::pack_convbn(const std::string& p) {
  ConvBN c;
  const int cin = (int)sd_.get(p + ".conv.weight").shape[1];
  const int mode = conv_mode(cin);
  c.conv = pack_conv(arena_, sd_, {p + ".conv"}, mode, mode == MODE_ROW ? pix_channels(cin) : 0);
  ChannelAffine a = bn_affine(sd_, p + ".bn");
  std::vector<float> sc(a.scale.begin(), a.scale.end()), sh(a.shift.begin(), a.shift.end());
  c.sc_off = pack_vector(arena_, sc);
  c.sh_off = pack_vector(arena_, sh);
  return c;
}

void VaeEncoder::finalize() {
  ATDN_CHECK(!ready_, "finalize called twice");
  stem_ = pack_convbn("encoder.0");
  for (int i = 0; i < 6; ++i) {
    const std::string p = "encoder." + std::to_string(i + 1);
    const int cin = kCh[i];
    ATDN_CHECK((int)sd_.get(p + ".conv.1.conv.weight").shape[0] == kCh[i + 1], "unexpected MappingVAE channel plan");
    res_[i].a = pack_convbn(p + ".conv.0");
    res_[i].b = pack_convbn(p + ".conv.1");
    const int mode = conv_mode(cin);
    res_[i].skip = pack_conv(arena_, sd_, {p + ".skip_layer"}, mode, mode == MODE_ROW ? pix_channels(cin) : 0);
    ChannelAffine a = bn_affine(sd_, p + ".out_block.1");
    std::vector<float> sc(a.scale.begin(), a.scale.end()), sh(a.shift.begin(), a.shift.end());
    res_[i].sc_off = pack_vector(arena_, sc);
    res_[i].sh_off = pack_vector(arena_, sh);
  }
  mean_ = pack_conv(arena_, sd_, {"mean_lin"}, MODE_TAP, 0);
  arena_.upload();
  auto fix = [&](ConvBN& c) { resolve(arena_, c.conv); c.sc = arena_.dev(c.sc_off); c.sh = arena_.dev(c.sh_off); };
  fix(stem_);
  for (auto& r : res_) { fix(r.a); fix(r.b); resolve(arena_, r.skip); r.sc = arena_.dev(r.sc_off); r.sh = arena_.dev(r.sh_off); }
  resolve(arena_, mean_);
  // every activation of the stack has at most H*W*4 floats per image (full-res maps carry 3 of 4 channels)
  const long cap = (long)maxB * H * W * 4;
  for (DeviceBuf* b : {&in4_, &bufA_, &bufB_, &bufS_}) {
    b->alloc(cap);
    ATDN_HIP(hipMemset(b->p, 0, (size_t)cap * sizeof(float)));  // the unused 4th channel must be finite
  }
  ready_ = true;
}

void VaeEncoder::encode(const float* images, int B, float* mu, hipStream_t st) {
  ATDN_CHECK(ready_, "weights not finalized");
  ATDN_CHECK(B >= 1 && B <= maxB, "batch exceeds max_batch of this handle");
  launch_prep_rgb(images, B, H, W, in4_.p, st);
  auto shape = [&](const PackedConv& L, const float* src, int h, int w, int stride, int pad) {
    ConvShape s;
    s.src0 = src; s.ld0 = L.C; s.sb0 = (long)h * w * L.C; s.C0 = L.C; s.H = h; s.W = w;
    s.KH = L.KH; s.KW = L.KW; s.stride = stride; s.padH = pad; s.padW = pad;
    s.w = L.w; s.ldw = L.ldw; s.N = L.N; s.nimg = B;
    return s;
  };
  int h = H, w = W;
  float* x = bufA_.p; float* t = bufB_.p;
  int ldx = 4;  // channels per pixel of x as the next layer reads it
  run_conv(MODE_ROW, shape(stem_.conv, in4_.p, h, w, 1, 3),
           EpiMishBN{stem_.conv.b, stem_.sc, stem_.sh, x, (long)h * w * ldx, ldx}, st);
  for (int i = 0; i < 6; ++i) {
    const Res& r = res_[i];
    const int cin = kCh[i], cout = kCh[i + 1];
    const int mode = conv_mode(cin);
    const int ldo = pix_channels(cout);
    const int oh = conv_out(h, 3, 2, 1), ow = conv_out(w, 3, 2, 1);
    ATDN_CHECK(r.a.conv.C == ldx && r.b.conv.C == ldx && r.skip.C == ldx, "channel layout mismatch between VAE layers");
    run_conv(mode, shape(r.a.conv, x, h, w, 1, 1), EpiMishBN{r.a.conv.b, r.a.sc, r.a.sh, t, (long)h * w * ldx, ldx}, st);
    run_conv(mode, shape(r.skip, x, h, w, 2, 0), EpiBias<ACT_NONE>{r.skip.b, bufS_.p, (long)oh * ow * ldo, ldo, 1.f}, st);
    // x is dead after the skip conv: the block output overwrites it
    run_conv(mode, shape(r.b.conv, t, h, w, 2, 1),
             EpiMishBNSkipMishBN{r.b.conv.b, r.b.sc, r.b.sh, bufS_.p, (long)oh * ow * ldo, ldo, r.sc, r.sh, x,
                                 (long)oh * ow * ldo, ldo}, st);
    h = oh; w = ow; ldx = ldo;
  }
  ATDN_CHECK(h == oh_ && w == ow_ && ldx == 128, "unexpected encoder output geometry");
  conv_dispatch<MODE_TAP>(shape(mean_, x, h, w, 1, 0), EpiBias<ACT_NONE>{mean_.b, mu, (long)h * w * 128, 128, 1.f}, st);
}

}  // namespace atdn
