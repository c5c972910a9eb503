// Host-side checkpoint staging and weight packing for the implicit-GEMM engine.
#pragma once
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "conv_mfma.h"

namespace atdn {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  long numel() const { long n = 1; for (auto d : shape) n *= d; return n; }
};

class StateDict {
 public:
  void put(const std::string& key, const float* data, const int64_t* shape, int rank) {
    std::string k = key;
    if (k.rfind("module.", 0) == 0) k = k.substr(7);  // DataParallel checkpoints (neural_slam.py:51-52)
    HostTensor t;
    t.shape.assign(shape, shape + rank);
    const long n = t.numel();
    t.data.assign(data, data + n);
    map_[k] = std::move(t);
  }
  const HostTensor& get(const std::string& key) const {
    auto it = map_.find(key);
    if (it == map_.end()) throw Error("missing state-dict entry: " + key);
    return it->second;
  }
  bool has(const std::string& key) const { return map_.count(key) != 0; }
  std::vector<std::string> keys() const {
    std::vector<std::string> k;
    for (auto& e : map_) k.push_back(e.first);
    return k;
  }
  size_t size() const { return map_.size(); }

 private:
  std::map<std::string, HostTensor> map_;
};

// A packed layer inside the device weight arena (offsets in floats until the arena is uploaded).
struct PackedConv {
  long w_off = -1, b_off = -1;
  int N = 0, ldw = 0, KH = 1, KW = 1, C = 0;  // C = (padded) channels per pixel the kernel will see
  int mode = MODE_TAP;
  float wscale = 1.f;  // sf packing: accumulator multiplier (power of two)
  const float* w = nullptr;
  const float* b = nullptr;
  // sf packing of 3x3 / 1x5 / 5x1 kernels: a second copy in MFMA-fragment order (conv_sf6.h)
  long wf_off = -1;
  const float* wf = nullptr;
  // ... and one in the operand order of v_mfma_f32_16x16x32_f16 (pack_fragment_major16)
  long wf16_off = -1;
  const float* wf16 = nullptr;
};

class WeightArena {
 public:
  long alloc(long n) {  // 16-byte aligned
    const long off = (long)host_.size();
    host_.resize(off + ((n + 3) / 4) * 4, 0.f);
    return off;
  }
  float* at(long off) { return host_.data() + off; }
  void upload() {
    ATDN_HIP(hipMalloc(&dev_, host_.size() * sizeof(float)));
    ATDN_HIP(hipMemcpy(dev_, host_.data(), host_.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  const float* dev(long off) const { return dev_ + off; }
  void release() { if (dev_) { (void)hipFree(dev_); dev_ = nullptr; } }
  size_t bytes() const { return host_.size() * sizeof(float); }

 private:
  std::vector<float> host_;
  float* dev_ = nullptr;
};

// Per-output-channel affine folded into a conv: w' = w*scale[n], b' = b*scale[n] + shift[n]
struct ChannelAffine { std::vector<double> scale, shift; };

inline ChannelAffine bn_affine(const StateDict& sd, const std::string& p, double eps = 1e-5) {
  const auto& w = sd.get(p + ".weight"); const auto& b = sd.get(p + ".bias");
  const auto& rm = sd.get(p + ".running_mean"); const auto& rv = sd.get(p + ".running_var");
  ChannelAffine a;
  const size_t n = w.data.size();
  a.scale.resize(n); a.shift.resize(n);
  for (size_t i = 0; i < n; ++i) {
    a.scale[i] = (double)w.data[i] / std::sqrt((double)rv.data[i] + eps);
    a.shift[i] = (double)b.data[i] - (double)rm.data[i] * a.scale[i];
  }
  return a;
}

// Pack torch-layout conv weights [Cout][Cin][KH][KW] (several tensors stacked along Cout) into K-contiguous rows.
//   TAP: row = [ky][kx][Cpad]           (Cpad = Cin rounded up to 32)
//   ROW: row = [ky][roundup(KW*Cpix,32)] with float index kx*Cpix + c  (Cpix = power of two >= Cin)
inline PackedConv pack_conv(WeightArena& A, const StateDict& sd, const std::vector<std::string>& names, int mode,
                            int Cpix, const ChannelAffine* fold = nullptr, bool has_bias = true) {
  PackedConv L;
  const HostTensor& w0 = sd.get(names[0] + ".weight");
  ATDN_CHECK(w0.shape.size() == 4, "conv weight must be 4-d");
  const int Cin = (int)w0.shape[1], KH = (int)w0.shape[2], KW = (int)w0.shape[3];
  L.mode = mode; L.KH = KH; L.KW = KW;
  L.C = (mode == MODE_TAP) ? round_up(Cin, 32) : Cpix;
  ATDN_CHECK(L.C >= Cin, "channel padding smaller than Cin");
  L.ldw = packed_k(mode, KH, KW, L.C);
  int N = 0;
  for (auto& nm : names) N += (int)sd.get(nm + ".weight").shape[0];
  L.N = N;
  L.w_off = A.alloc((long)N * L.ldw);
  L.b_off = A.alloc(N);
  int n0 = 0;
  for (auto& nm : names) {
    const HostTensor& w = sd.get(nm + ".weight");
    ATDN_CHECK((int)w.shape[1] == Cin && (int)w.shape[2] == KH && (int)w.shape[3] == KW, "stacked convs must agree");
    const int Co = (int)w.shape[0];
    const float* bias = (has_bias && sd.has(nm + ".bias")) ? sd.get(nm + ".bias").data.data() : nullptr;
    for (int n = 0; n < Co; ++n) {
      const double sc = fold ? fold->scale[n0 + n] : 1.0;
      float* row = A.at(L.w_off + (long)(n0 + n) * L.ldw);
      for (int c = 0; c < Cin; ++c)
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < KW; ++kx) {
            const double v = (double)w.data[(((long)n * Cin + c) * KH + ky) * KW + kx] * sc;
            const long k = (mode == MODE_TAP) ? ((long)(ky * KW + kx) * L.C + c)
                                              : ((long)ky * round_up(KW * L.C, 32) + kx * L.C + c);
            row[k] = (float)v;
          }
      double bv = bias ? (double)bias[n] : 0.0;
      if (fold) bv = bv * sc + fold->shift[n0 + n];
      A.at(L.b_off)[n0 + n] = (float)bv;
    }
    n0 += Co;
  }
  return L;
}

// Fragment-major copy of an sf-packed weight matrix for the 16x16x32 loop of conv_sf6.h: [ceil(N/16)][K/32][hi, lo][lane] x 16 B. Lane (r = lane & 15,
// g = lane >> 4) of the wave that multiplies output channels nb*16 .. nb*16+15 finds, for K chunk q, its hi operand (16-byte
// slot g of row nb*16+r, chunk q: halves 8g .. 8g+7) at ((nb*nq + q)*2 + 0)*1024 + lane*16 bytes and its lo operand (slot
// 4+g) 1024 bytes further: every wave load is one contiguous KiB. Rows >= N are zero.
inline void pack_fragment_major16(WeightArena& A, PackedConv& L) {
  const int nq = L.ldw / 32, nblk = (L.N + 15) / 16;
  L.wf16_off = A.alloc((long)nblk * nq * 512);
  const float* w = A.at(L.w_off);  // (alloc may have moved the arena: take the pointers after it)
  float* f = A.at(L.wf16_off);
  for (int nb = 0; nb < nblk; ++nb)
    for (int q = 0; q < nq; ++q)
      for (int hl = 0; hl < 2; ++hl)
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 15, gq = lane >> 4, row = nb * 16 + r, slot = 4 * hl + gq;
          float* d = f + ((((long)nb * nq + q) * 2 + hl) * 64 + lane) * 4;
          if (row < L.N) std::memcpy(d, w + (long)row * L.ldw + q * 32 + slot * 4, 16);
          else std::memset(d, 0, 16);
        }
}

// Same as pack_conv(TAP) but in split-f16 form (sf.h): every 32-float K-chunk of a row becomes [32 hi | 32 lo]
// halves; all weights of the layer are pre-multiplied by 2^p so that max|w| lands in [1,2) and L.wscale = 2^-p.
inline PackedConv pack_conv_sf(WeightArena& A, const StateDict& sd, const std::vector<std::string>& names,
                               const ChannelAffine* fold = nullptr, bool has_bias = true) {
  PackedConv L = pack_conv(A, sd, names, MODE_TAP, 0, fold, has_bias);  // fp32 rows first (BN already folded)
  float* w = A.at(L.w_off);
  const long total = (long)L.N * L.ldw;
  float mx = 0.f;
  for (long i = 0; i < total; ++i) mx = std::max(mx, std::fabs(w[i]));
  int e = 0;
  if (mx > 0.f) (void)std::frexp(mx, &e);  // mx = f * 2^e, f in [0.5,1)
  const int p = 1 - e;                      // w * 2^p has its maximum in [1,2)
  L.wscale = std::ldexp(1.0f, -p);
  std::vector<float> row(L.ldw);
  for (int n = 0; n < L.N; ++n) {
    float* r = w + (long)n * L.ldw;
    std::memcpy(row.data(), r, L.ldw * sizeof(float));
    _Float16* hrow = reinterpret_cast<_Float16*>(r);
    for (int q = 0; q < L.ldw / 32; ++q)
      for (int j = 0; j < 32; ++j) {
        const float v = std::ldexp(row[q * 32 + j], p);
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        hrow[q * 64 + j] = hi;
        hrow[q * 64 + 32 + j] = lo;
      }
  }
  if ((L.KH == 3 && L.KW == 3) || (L.KH == 1 && L.KW == 5) || (L.KH == 5 && L.KW == 1)) {
    pack_fragment_major16(A, L);
  }
  return L;
}

// Split-f16 fragment-major copy of the 7x7 stem (ROW-packed fp32 rows [64][7 x 32], k = ky*32 + kx*4 + c) for
// stem_sf.hip: [n-tile j = 0,1][K step q = 0..13][hi, lo][lane] x 16 B; lane (r, h) holds row 32 j + r, k = 16 q + 8 h .. + 7.
// Weights are pre-multiplied by 2^p (maximum in [1,2)); L.wscale = 2^-p.
inline void pack_stem_sf(WeightArena& A, PackedConv& L) {
  ATDN_CHECK(L.mode == MODE_ROW && L.KH == 7 && L.KW == 7 && L.C == 4 && L.ldw == 224 && L.N == 64, "stem shape");
  L.wf_off = A.alloc(2L * 14 * 2 * 256);
  const float* w = A.at(L.w_off);  // (alloc may have moved the arena)
  float mx = 0.f;
  for (long i = 0; i < (long)L.N * L.ldw; ++i) mx = std::max(mx, std::fabs(w[i]));
  int e = 0;
  if (mx > 0.f) (void)std::frexp(mx, &e);
  const int p = 1 - e;
  L.wscale = std::ldexp(1.0f, -p);
  _Float16* f = reinterpret_cast<_Float16*>(A.at(L.wf_off));
  for (int nt = 0; nt < 2; ++nt)
    for (int q = 0; q < 14; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        for (int m = 0; m < 8; ++m) {
          const float v = std::ldexp(w[(long)(nt * 32 + r) * L.ldw + 16 * q + 8 * h + m], p);
          const _Float16 hi = (_Float16)v;
          f[(((long)(nt * 14 + q) * 2 + 0) * 64 + lane) * 8 + m] = hi;
          f[(((long)(nt * 14 + q) * 2 + 1) * 64 + lane) * 8 + m] = (_Float16)(v - (float)hi);
        }
      }
}

// Stack several convs along Cout, keep only input channels in the given [begin,end) ranges (in that order), and pack
// the result in sf form. Used to split the ConvGRU weights into the iteration-invariant context part and the rest.
inline PackedConv pack_conv_sf_channels(WeightArena& A, const StateDict& sd, const std::vector<std::string>& names,
                                        const std::vector<std::pair<int, int>>& ranges, bool has_bias) {
  StateDict tmp;
  std::vector<std::string> tnames;
  for (size_t t = 0; t < names.size(); ++t) {
    const HostTensor& w = sd.get(names[t] + ".weight");
    const int Co = (int)w.shape[0], Ci = (int)w.shape[1], KH = (int)w.shape[2], KW = (int)w.shape[3];
    int Cn = 0;
    for (auto& r : ranges) Cn += r.second - r.first;
    std::vector<float> out((size_t)Co * Cn * KH * KW);
    for (int n = 0; n < Co; ++n) {
      int c2 = 0;
      for (auto& r : ranges)
        for (int c = r.first; c < r.second; ++c, ++c2)
          std::memcpy(&out[((size_t)n * Cn + c2) * KH * KW], &w.data[((size_t)n * Ci + c) * KH * KW],
                      sizeof(float) * KH * KW);
    }
    const std::string key = "s" + std::to_string(t);
    const int64_t shp[4] = {Co, Cn, KH, KW};
    tmp.put(key + ".weight", out.data(), shp, 4);
    if (has_bias && sd.has(names[t] + ".bias")) {
      const HostTensor& b = sd.get(names[t] + ".bias");
      const int64_t bs[1] = {Co};
      tmp.put(key + ".bias", b.data.data(), bs, 1);
    }
    tnames.push_back(key);
  }
  return pack_conv_sf(A, tmp, tnames, nullptr, has_bias);
}

inline long pack_vector(WeightArena& A, const std::vector<float>& v) {
  const long off = A.alloc((long)v.size());
  std::memcpy(A.at(off), v.data(), v.size() * sizeof(float));
  return off;
}

inline void resolve(const WeightArena& A, PackedConv& L) {
  L.w = A.dev(L.w_off);
  L.b = A.dev(L.b_off);
  L.wf = L.wf_off >= 0 ? A.dev(L.wf_off) : nullptr;
  L.wf16 = L.wf16_off >= 0 ? A.dev(L.wf16_off) : nullptr;
}

}  // namespace atdn
