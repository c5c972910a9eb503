// Split-f16 halo-patch convolution: an M tile is a TH x TW block of OUTPUT pixels of one image; per 32-channel
// chunk the (TH+KH-1) x (TW+KW-1) input patch (true zero padding included) is staged in LDS ONCE and all KH*KW taps
// read it at an address offset. Weights never touch LDS: earlier generations (round 1, since removed) streamed a
// [BN][32-chunk] weight tile through LDS every (chunk, tap) step and paid one workgroup barrier per step for it;
// with two waves per SIMD that lock-step held the matrix pipe at ~40 % busy whichever way the tile reached LDS.
// Here every wave loads ITS OWN weight fragments straight from global memory (L2/L1-resident: the whole layer is
// 1-2 MB) into registers in MFMA operand layout, one step ahead of use:
//   lane (r = lane & 31, h = lane >> 5) holds, for output channel n0 + (wn*TN + j)*32 + r and K sub-step t,
//   the 16-byte slots 2t+h (hi) and 4+2t+h (lo) of that row's 128-byte [32 hi | 32 lo] chunk.
// Only the activation patch lives in LDS, so the only barriers are the two around its refresh at a chunk boundary
// (every KH*KW steps); between them the waves of a block run free and overlap each other's LDS reads and MFMAs.
// The wave grid is WM x WN over (TH*TW pixels) x (BN channels); WN = BN/32 makes the weight loads non-redundant.
// KH x KW is a template parameter: the tap loop is unrolled, tap offsets are immediates and wait counts exact.
//
// Epilogues that provide the channel-vector form (`kVec4`: store4 / load4 / apply4 on 4 consecutive channels of
// one pixel) run with the MFMA operands swapped — weights as the row operand — so a lane ends up with 16 channels
// of ONE pixel (4 runs of 4): its epilogue traffic is 8- and 16-byte accesses instead of 2-byte ones, 4x fewer
// VMEM instructions. Epilogues without it (InstanceNorm statistics need the pixel-major registers) keep the
// classic orientation.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "conv_sf.h"

namespace atdn {

struct Conv2Geom {
  const float* src0; const float* src1;
  long sb0, sb1;
  int ld0, ld1, C0, C1;
  int H, W, Ho, Wo;
  int KH, KW, padH, padW;
  int PH, PW;            // patch size in pixels
  int tiles_x, tiles_y;  // per image
  int nimg, ntile_n;
  const float* w; int ldw; int N;
  float wscale;
  // normalise-on-load (NORM): src0 is RAW fp32 [pix][C0] and the patch loader applies
  // relu((x - in_mean[img][c]) * in_rstd[img][c]) before splitting to sf (InstanceNorm + ReLU of the producer)
  const float* in_mean; const float* in_rstd;
};

// shapes the halo-patch kernel serves: stride-1 3x3 / 1x5 / 5x1 convolutions with shared weights
inline bool conv_halo_eligible(const ConvShape& s) {
  if (s.stride != 1 || s.wb != 0) return false;
  return (s.KH == 3 && s.KW == 3) || (s.KH == 1 && s.KW == 5) || (s.KH == 5 && s.KW == 1);
}

// which kernel shapes an epilogue is instantiated for on this path (bit 0: 3x3, bit 1: 1x5 and 5x1)
template <class E, class = void> struct epi_gen6 : std::integral_constant<int, 0> {};
template <class E> struct epi_gen6<E, std::void_t<decltype(E::kGen6)>> : std::integral_constant<int, E::kGen6> {};
template <class E, class = void> struct EpiAux4 { struct type {}; };
template <class E> struct EpiAux4<E, std::void_t<typename E::Aux4>> { using type = typename E::Aux4; };
template <class E, class = void> struct epi_flowhead : std::false_type {};
template <class E> struct epi_flowhead<E, std::void_t<decltype(E::kFlowHead)>> : std::bool_constant<E::kFlowHead> {};
template <class E, class = void> struct epi_ring3 : std::false_type {};
template <class E> struct epi_ring3<E, std::void_t<decltype(E::kRing3)>> : std::bool_constant<E::kRing3> {};
template <class E, class = void> struct epi_rawacc : std::false_type {};
template <class E> struct epi_rawacc<E, std::void_t<decltype(E::kRawAcc)>> : std::bool_constant<E::kRawAcc> {};
template <class E, class = void> struct epi_vec4 : std::false_type {};
template <class E> struct epi_vec4<E, std::void_t<decltype(E::kVec4)>> : std::bool_constant<E::kVec4> {};

// FAST: plain-f16 arithmetic (precision mode 2): only the hi x hi MFMA of every product is issued
// NORM: normalise-on-load. src0 holds the RAW fp32 output of an InstanceNorm'ed conv (same addressing: 128 bytes per
// pixel and 32-channel chunk); the patch loader applies relu((x - mean) * rstd) per (image, channel), splits to hi | lo and
// writes the sf chunk image itself, one patch row per half-step over the last NP half-steps of a chunk — the separate
// normalisation pass between the two convs of a residual block (read 4 B + write 4 B per element) disappears.
// The loop runs on v_mfma_f32_16x16x32_f16 (K = 32 per instruction: a whole 32-channel chunk of one tap). Same FLOP per cycle
// and operand bytes per FLOP as v_mfma_f32_32x32x16_f16 (the loop rounds 1-2 ran, removed in round 4 after a round of A/B
// tests), but on random data the chip holds a higher clock under this shape: the conv-like loop of tools/microbench (pixel
// operands from LDS, weights in registers, two waves per SIMD) runs at 1914 instead of 1575 TF/s executed
// (profiles/r03_microbench_mfma.txt; equal on all-zero operands: it is the clock, MI355X_MICROARCH.md "DVFS give-back" 7):
//   * operand lane map: lane (n = lane & 15, g = lane >> 4) holds, for row / column n of a 16-wide block, the 16-byte slot
//     g of the chunk's [32 hi] (or [32 lo]) halves; a wave's 32-pixel row tile is two 16-pixel blocks = the two patch rows
//     it covers (TW = 16), its 32-channel tile two 16-channel blocks;
//   * the pixel pitch of the patch image is 160 B (144 B puts slot g + 1 of pixels 4-11 on the banks of slot g of pixels
//     12-15 and 0-3 in the lane groups of ds_read_b128; 160 B = 10 slots is conflict-free for this lane map);
//   * weights come from the fragment-major copy (weights.h: pack_fragment_major16): [N/16][K/32][hi | lo][lane] x 16 B, one
//     contiguous KiB per wave load;
//   * a K step (tap, chunk) is two half-steps — pixel half 0 (first patch row of every row tile) and half 1 — with the
//     pixel fragments of the next half-step read during the current one; the weights of a K step (channel block 0 loaded in
//     half-step 0, block 1 in half-step 1) stay in a ring of RT K-step slots, RT - 1 steps ahead;
//   * accumulator: b[pixel half][channel block] of 4 registers: in SWAP mode (weights = row operand) lane (n, g) holds
//     channels 16 cb + 4 g + 0..3 of pixel 16 half + n.
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct SfAcc { f32x4v b[2][2]; };

// Diagnostic builds only (python -m atdn_vslam_amd.build --variant stamp -DATDN_CONV_STAMP; tools/diag/conv_stamps.py): where
// the life of a block of the ConvGRU kernels goes. Wave 0 of every block records s_memtime at its start, after the prologue
// (first patch published), at the end of the main loop and at the end of the epilogue, the cycles it waited at the chunk
// barriers, the 100 MHz real-time clock at both ends and the hardware id of its CU. The product build compiles none of it.
#ifdef ATDN_CONV_STAMP
#define ATDN_CONV_STAMP_SLOTS 4096
extern __device__ unsigned long long atdn_conv_stamps_dev[4][ATDN_CONV_STAMP_SLOTS][12];
template <class E> struct conv_stamp_kind : std::integral_constant<int, -1> {};
#endif

// ABL (tools/microbench only — diagnostic builds with WRONG results, timing only; the product instantiates ABL = 0): bit 0 no
// weight loads in the loop, bit 1 no patch refresh (and no chunk-boundary barrier), bit 2 no LDS fragment reads in the loop,
// bit 3 no epilogue, bit 4 every block stores its tile to the SAME 192 pixels of image 0 (the epilogue's instructions all run,
// the stores stay in L2: what the HBM write side of the epilogue costs)
// (the body: one block of one convolution. `bidx` is the block's index within ITS convolution — blockIdx.x in conv_sf6_kernel, or
// the index behind the first convolution's blocks in conv_sf6_pair_kernel, which runs two independent convolutions of the same
// shape class as ONE launch)
template <int TH, int TW, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST = false, bool NORM = false, int ABL = 0>
__device__ __forceinline__ void conv_sf6_body(const Conv2Geom& g, const Epi& ep, const int bidx) {
  static_assert(ABL == 0 || !NORM, "ablation builds exist for the plain patch loader");
  static_assert(TW == 16, "the 16x16x32 loop is built for 16-pixel tile rows");
  constexpr int NIMG = 2;   // two patch images: a chunk boundary costs one barrier
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int TM = TH * TW / 32 / WM, TN = BN / 32 / WN;
  static_assert(TM * WM * 32 == TH * TW && TN * WN * 32 == BN, "wave grid must tile the block");
  static_assert(32 % TW == 0 || TW % 32 == 0, "a 32-pixel MFMA row tile must cover whole tile rows");
  constexpr int NTAP = KH * KW;
  constexpr int RSTEP = NT / 8;
  constexpr int PROWS = (TH + KH - 1) * (TW + KW - 1);
  constexpr int PW = TW + KW - 1;
  constexpr int NP = (PROWS + RSTEP - 1) / RSTEP;
  constexpr int ROWB = 160;   // pixel pitch of the patch image
  // LDS pitch between patch rows: a fragment read stays inside one patch row, any pitch works
  constexpr int RS = PW * ROWB;
  constexpr bool SWAP = epi_vec4<Epi>::value;
  // two patch images: chunk c+1 is written (from the registers its loads landed in) during the last tap of chunk c,
  // so a chunk boundary costs one barrier, not two
  constexpr int PSZ = (TH + KH - 1) * RS;
  __shared__ __attribute__((aligned(256))) char Pbytes[NIMG * PSZ];

  const int tid = threadIdx.x;
  const int tiles_img = g.tiles_x * g.tiles_y;
  const int nblk = g.nimg * tiles_img * g.ntile_n;
  const int id = xcd_remap(bidx, nblk);
  const int tile_n = id % g.ntile_n;
  const int tmg = id / g.ntile_n;
  const int img = tmg / tiles_img;
  const int tloc = tmg - img * tiles_img;
  const int ty0 = (tloc / g.tiles_x) * TH, tx0 = (tloc % g.tiles_x) * TW;
  const int n0 = tile_n * BN;
  const int lane = tid & 63, wave = tid >> 6;
#ifdef ATDN_CONV_STAMP
  constexpr int kStamp = conv_stamp_kind<Epi>::value;
  unsigned long long st_r0 = 0, st_t0 = 0, st_t1 = 0, st_t2 = 0, st_bar = 0;
  unsigned long long st_e_slab = 0, st_e_wait = 0, st_e_apply = 0;   // epilogue: transpose through the slab / operand wait / arithmetic + stores
  if constexpr (kStamp >= 0) { st_r0 = __builtin_amdgcn_s_memrealtime(); st_t0 = __builtin_amdgcn_s_memtime(); }
#endif

  // ---- patch loader role (registers, true zero padding)
  // (NORM: its loader writes 8-byte halves, ds_write_b64 = groups of 16 lanes = two patch rows; with rows 4 apart —
  // 4 x 36 dwords = 16 mod 32 banks — instead of adjacent the two rows' 16 banks do not overlap. SQ_LDS_BANK_CONFLICT of
  // the normalise-on-load kernels was 11-15 % of their LDS cycles.)
  // (pixel pitch 160 B = 40 dwords: rows TWO apart are 16 banks apart)
  const int jrow = (tid >> 3) & 7;
  const int s = tid & 7, r0 = !NORM ? tid >> 3 : ((tid >> 6) << 3) + 2 * (jrow & 1) + ((jrow >> 1) & 1) + 4 * (jrow >> 2);
  // per patch row of this thread: (pixel offset in the image + 1, 0 = zero padding) << 12 | LDS offset / 16
  static_assert((TH + KH - 1) * RS / 16 <= 4096, "LDS offset field");
  unsigned pmeta[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int prow = r0 + RSTEP * k;
    unsigned off = 0;
    const int py = prow / PW, px = prow - py * PW;
    if (prow < PROWS) {
      const int iy = ty0 - g.padH + py, ix = tx0 - g.padW + px;
      if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) off = (unsigned)(iy * g.W + ix) + 1u;
    }
    pmeta[k] = (off << 12) | (unsigned)((py * RS + px * ROWB + 16 * s) >> 4);
    // a row past the patch repeats this thread's first row (same data to the same address): the refresh stays
    // branch-free, so the whole chunk body is one scheduling region
    if (prow >= PROWS) pmeta[k] = pmeta[0];
  }
  const float* s0 = g.src0 + (long)img * g.sb0;
  const float* s1 = g.src1 ? g.src1 + (long)img * g.sb1 : nullptr;
  const int nck = (g.C0 + g.C1) >> 5;

  float4 pr[NP];
  auto fetch_patch = [&](int c) {
    const int cc = c << 5;
    const float* sp; int ld, co;
    if (cc < g.C0) { sp = s0; ld = g.ld0; co = cc; } else { sp = s1; ld = g.ld1; co = cc - g.C0; }
#pragma unroll
    for (int k = 0; k < NP; ++k)
      pr[k] = *reinterpret_cast<const float4*>(sp + (long)((pmeta[k] >> 12) ? (pmeta[k] >> 12) - 1 : 0) * ld + co + 4 * s);
  };
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NP; ++k)
      *reinterpret_cast<float4*>(Pbytes + buf * PSZ + ((pmeta[k] & 0xFFFu) << 4)) = keep_if((pmeta[k] >> 12) != 0, pr[k]);
  };
  // NORM: per-channel constants of the chunk being fetched (this thread's 4 channels), and the store of ONE patch row
  float4 nmu = make_float4(0.f, 0.f, 0.f, 0.f), nrs = make_float4(1.f, 1.f, 1.f, 1.f);
  auto fetch_norm = [&](int c) {
    nmu = *reinterpret_cast<const float4*>(g.in_mean + (long)img * g.C0 + (c << 5) + 4 * s);
    nrs = *reinterpret_cast<const float4*>(g.in_rstd + (long)img * g.C0 + (c << 5) + 4 * s);
  };
  // (round 5: the split of the four normalised values as two pair conversions + four v_fma_mix residuals with the saturation
  // flag in a register — sf.h — instead of four counted sf_split calls, each with a compare, a two-instruction clamp, three
  // conversions, a subtraction and a cold branch to the device counter: the normalise-on-load loop carried 465 vector
  // instructions per 324 MFMAs against 151 in the plain loader)
  bool norm_sat = false;
  auto store_row_norm = [&](int buf, int k, float4 mu, float4 rs) {
    // ReLU, the format's clamp and the zero padding in ONE v_med3 per value: the upper limit is 65504 for a pixel of the image
    // and 0 for a padding pixel (whose registers hold some other pixel's data)
    const float hi = (pmeta[k] >> 12) != 0 ? 65504.f : 0.f;
    float4 v = pr[k];
    v.x = (v.x - mu.x) * rs.x; v.y = (v.y - mu.y) * rs.y; v.z = (v.z - mu.z) * rs.z; v.w = (v.w - mu.w) * rs.w;
    // (a NaN of the raw tensor is dropped by the maximum below and by v_med3 and shows as 0 here. It cannot pass unseen: one NaN
    // or infinity in a channel makes that channel's mean / rstd non-finite, and in_finalize_merge_kernel (kernels.hip) raises the
    // saturation alarm for non-finite statistics — the producers store raw fp32 and raise none of their own. Two unordered tests
    // per patch row here would be 16 more vector instructions per 324 MFMAs of the hottest loop of the feature network.)
    const float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    norm_sat |= !(m <= 65504.f);
    v.x = __builtin_amdgcn_fmed3f(v.x, 0.f, hi); v.y = __builtin_amdgcn_fmed3f(v.y, 0.f, hi);
    v.z = __builtin_amdgcn_fmed3f(v.z, 0.f, hi); v.w = __builtin_amdgcn_fmed3f(v.w, 0.f, hi);
    const unsigned h0 = sf_cvt_pk_(v.x, v.y), h1 = sf_cvt_pk_(v.z, v.w);
    const unsigned l0 = sf_residual_pk_(h0, v.x, v.y), l1 = sf_residual_pk_(h1, v.z, v.w);
    // the 16-byte slot field addresses slot s of the pixel: its 128-byte chunk starts s slots earlier
    char* px = Pbytes + buf * PSZ + (((pmeta[k] & 0xFFFu) - (unsigned)s) << 4);
    *reinterpret_cast<uint2*>(px + 8 * s) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(px + 64 + 8 * s) = make_uint2(l0, l1);
  };

  // ---- MFMA roles
  const int wm = wave / WN, wn = wave % WN;
  SfAcc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j].b[e >> 3][(e >> 2) & 1][e & 3] = 0.f;
    }
  {
    const int n16 = lane & 15, g16 = lane >> 4;
    int a_off[TM];   // first patch row (pixel half 0) of row tile i: pixel n16, slot g16; half 1 is the next patch row
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = ((wm * TM + i) * 2) * RS + n16 * ROWB + 16 * g16;
    const char* Pb = Pbytes;
    // weight fragments of channel block cb of column tile j: [n/16][q][hi | lo][lane] x 16 B
    const int nblk16 = (g.N + 15) >> 4;
    const float* wrow[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        wrow[j][cb] = g.w + (long)min(((n0 >> 4) + (wn * TN + j) * 2 + cb), nblk16 - 1) * (NTAP * nck) * 512 + lane * 4;
    // ring of RT K-step slots; the chunk loop is unrolled over CU chunks so that every slot index is a compile-time constant
    // (three slots for the 5-tap ConvGRU gates too — their 12 chunks are whole iterations of the 3-chunk unroll that a 3-slot ring
    // over 5 taps needs: q gates -2.4 %, z|r -0.3 %; the context convolutions, 4 chunks, would multiply two surplus chunks and stay
    // on two slots. profiles/r05_ab_gru_ring3.txt)
    constexpr int RT = (NTAP % 3 == 0 || epi_ring3<Epi>::value) ? 3 : 2;
    constexpr int CU = (NTAP % RT == 0) ? 1 : RT;
    constexpr int NS = CU * NTAP;            // K steps per iteration of the (unrolled) chunk loop
    static_assert(NS % RT == 0, "ring size must divide the K steps of an unrolled iteration");
    struct WFrag { f16x8 hi[TN][2], lo[TN][2]; };
    WFrag wr[RT];
    // K step `st` (0 .. NS-1 of the iteration that starts at chunk c0; st >= NS: the next iteration): chunk and tap
    auto load_w = [&](WFrag& f, int c0, int st, int cb) __attribute__((always_inline)) {
      const int cc = min(c0 + st / NTAP, nck - 1);   // clamped: the prefetches behind the last chunk fetch valid, unused data
      const int q = (st % NTAP) * nck + cc;          // packed K order is [tap][channel chunk]
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f.hi[j][cb] = *reinterpret_cast<const f16x8*>(wrow[j][cb] + (long)q * 512);
        if constexpr (!FAST) f.lo[j][cb] = *reinterpret_cast<const f16x8*>(wrow[j][cb] + (long)q * 512 + 256);
      }
    };
    fetch_patch(0);
    if constexpr (NORM) fetch_norm(0);
#pragma unroll
    for (int st = 0; st < RT - 1; ++st) { load_w(wr[st], 0, st, 0); load_w(wr[st], 0, st, 1); }
    if constexpr ((ABL & 1) != 0) { load_w(wr[RT - 1], 0, RT - 1, 0); load_w(wr[RT - 1], 0, RT - 1, 1); }
    if constexpr (NORM) {
#pragma unroll
      for (int k = 0; k < NP; ++k) store_row_norm(0, k, nmu, nrs);
    } else {
      store_patch(0);
    }
    __syncthreads();
#ifdef ATDN_CONV_STAMP
    if constexpr (kStamp >= 0) st_t1 = __builtin_amdgcn_s_memtime();
#endif
    f16x8 ah[2][TM], al[2][TM];   // [pixel half][row tile]
    auto read_a = [&](int half, int tap) __attribute__((always_inline)) {
      const char* arow = Pb + (tap / KW + half) * RS + (tap % KW) * ROWB;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[half][i] = *reinterpret_cast<const f16x8*>(arow + a_off[i]);
        if constexpr (!FAST) al[half][i] = *reinterpret_cast<const f16x8*>(arow + a_off[i] + 64);
      }
    };
    auto mfma_half = [&](int half, const WFrag& w) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x4v c = acc[i][j].b[half][cb];
            if constexpr (SWAP) {
              if constexpr (!FAST) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi[j][cb], al[half][i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.lo[j][cb], ah[half][i], c, 0, 0, 0);
              }
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi[j][cb], ah[half][i], c, 0, 0, 0);
            } else {
              if constexpr (!FAST) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[half][i], w.hi[j][cb], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[half][i], w.lo[j][cb], c, 0, 0, 0);
              }
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[half][i], w.hi[j][cb], c, 0, 0, 0);
            }
            acc[i][j].b[half][cb] = c;
          }
    };
    constexpr int NMF = (FAST ? 1 : 3) * TM * TN * 2;   // MFMAs per half-step
    constexpr int NH = 2 * NTAP;                        // half-steps per chunk
    static_assert(!NORM || NH - 1 - NP >= 1, "normalise-on-load spreads its NP patch rows over the half-steps of a chunk");
    read_a(0, 0);
    // (round 5: every kernel but the instance-norm ones raises its waves' priority for the main loop and drops it for the
    // epilogue — a wave that is multiplying wins the issue slot over its SIMD partner's epilogue instructions: z|r -0.9 %,
    // q -1.8 %, motion encoder -1.2 %; on the statistics kernels of fnet the same cost 1 %, so they stay at the default; level 3
    // measures the same as level 2. profiles/r05_ab_setprio.txt)
    if constexpr (KH != 3 || !Epi::kStats) __builtin_amdgcn_s_setprio(2);
    for (int c0 = 0; c0 < nck; c0 += CU) {
#pragma unroll
      for (int cu = 0; cu < CU; ++cu) {
        const int c = c0 + cu;
        // (CU > 1 and an odd chunk count: the surplus chunk multiplies nothing but still fetches, stores and joins the
        // barrier — up to 1 / nck of the kernel; the ConvGRU shapes have nck = 12 or 16. ADVICE r3)
        const bool live = CU == 1 || c < nck;
        const int cn = min(c + 1, nck - 1);
#pragma unroll
        for (int hs = 0; hs < NH; ++hs) {
          const int tap = hs >> 1, half = hs & 1;
          const int st = cu * NTAP + tap;     // compile-time K step index inside the unrolled iteration
          __builtin_amdgcn_sched_barrier(0);
          if (!(ABL & 2) && hs == 0) { fetch_patch(cn); if constexpr (NORM) fetch_norm(cn); }   // lands during this chunk's taps
          // weights of K step st + RT - 1: channel block `half` in this half-step, into the slot K step st - 1 released
          // two-slot ring (1x5 / 5x1): both channel blocks of K step st + 1 in half-step 0 — block 1 loaded in half-step 1 would
          // be needed one half-step later (both blocks are multiplied in every half-step): q gate -1.5 %, z|r -0.6 % (round 4)
          if (!(ABL & 1)) {
            if (RT == 2) { if (half == 0) { load_w(wr[(st + 1) % RT], c0, st + 1, 0); load_w(wr[(st + 1) % RT], c0, st + 1, 1); } }
            else load_w(wr[(st + RT - 1) % RT], c0, st + RT - 1, half);
          }
          if (!(ABL & 4) && hs + 1 < NH) read_a((hs + 1) & 1, (hs + 1) >> 1);
          if constexpr (NORM) {
            if (hs >= NH - 1 - NP && hs <= NH - 2) store_row_norm((c + 1) & 1, hs - (NH - 1 - NP), nmu, nrs);
          } else {
            if (!(ABL & 2) && hs == NH - 2) store_patch((c + 1) & 1);   // that image was last read in chunk c - 1
          }
          if (live) mfma_half(half, wr[st % RT]);
          {
            constexpr int nds = 2 * TM;
            const int nvm = (RT == 2 ? (half == 0 ? 4 * TN : 0) : 2 * TN) + (hs == 0 ? NP : 0);
            const int ndw = NORM ? ((hs >= NH - 1 - NP && hs <= NH - 2) ? 2 : 0) : ((hs == NH - 2) ? NP : 0);
#pragma unroll
            for (int k = 0; k < NMF; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (hs + 1 < NH && k < nds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              else if (k - ((hs + 1 < NH) ? nds : 0) < nvm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              else if (k - ((hs + 1 < NH) ? nds : 0) - nvm < ndw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
          }
        }
        // chunk boundary: publish the next patch image (one barrier)
        if (!(ABL & 2)) {
#ifdef ATDN_CONV_STAMP
          unsigned long long st_b0 = 0;
          if constexpr (kStamp >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_b0 = __builtin_amdgcn_s_memtime(); }
#endif
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ATDN_CONV_STAMP
          if constexpr (kStamp >= 0) st_bar += __builtin_amdgcn_s_memtime() - st_b0;
#endif
          Pb = Pbytes + ((c + 1) & 1) * PSZ;
        }
        if (!(ABL & 4)) read_a(0, 0);
      }
    }
  }

#ifdef ATDN_CONV_STAMP
  if constexpr (kStamp >= 0) st_t2 = __builtin_amdgcn_s_memtime();
  auto stamp_out = [&]() __attribute__((always_inline)) {
    if constexpr (kStamp >= 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the epilogue's stores have left the wave
      const unsigned long long t3 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      if (wave == 0 && lane == 0 && bidx < ATDN_CONV_STAMP_SLOTS) {
        unsigned long long* o = atdn_conv_stamps_dev[kStamp * 2 + (KH == 5 ? 1 : 0)][bidx];
        o[0] = st_r0; o[1] = r1; o[2] = st_t1 - st_t0; o[3] = st_t2 - st_t1; o[4] = st_bar; o[5] = t3 - st_t2;
        o[6] = ((unsigned long long)xcc << 32) | hw; o[7] = t3 - st_t0;
        o[8] = st_e_slab; o[9] = st_e_wait; o[10] = st_e_apply; o[11] = 0;
      }
    }
  };
#endif
  if constexpr (KH != 3 || !Epi::kStats) __builtin_amdgcn_s_setprio(0);
  if constexpr (NORM) sf_report(norm_sat);
  if constexpr ((ABL & 8) != 0) {  // diagnostic: no epilogue (one conditional store keeps the accumulators alive)
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot += acc[i][j].b[e >> 3][(e >> 2) & 1][e & 3];
    if (tot == 1.2345e-30f) ep(img, 0, 0, tot);
    return;
  }
  if constexpr (SWAP) {
    // ---- channel-vector epilogue. After the MFMAs lane (n, g) holds, for pixel 16 half + n of a 32-pixel tile, channels
    // 16 cb + 4 g + (0..3). Storing from there gives every 128-byte output line 4 separate 32-byte partial writes (and the
    // operand loads of the gate / residual epilogues the same shape): measured in round 4 against the form below, plain-store
    // epilogues do not care but the ConvGRU gates and the residual add run 8-9 % slower (profiles/r04_ab_direct_epilogue.txt).
    // So each tile is transposed through a wave-private LDS slab first: 4 ds_write_b128 in, 4 ds_read_b128 out, after which
    // lane l owns pixel 8q + l/8, channels 4*(l%8)..+3 — 8 consecutive lanes cover one pixel's 32-channel group, and every
    // load and store of the epilogue (operands of the GRU gates included) is a whole line per pixel.
    // The slabs reuse the patch images (dead after the main loop; one barrier before the first write), so a block's
    // LDS footprint is the two patch images only and narrower blocks fit several to a CU: one block's epilogue then
    // overlaps another block's main loop.
    static_assert(NW * 32 * LDS_LD * 4 <= NIMG * PSZ, "transpose slabs must fit in the patch images");
    __syncthreads();
    float* tb = reinterpret_cast<float*>(Pbytes) + wave * (32 * LDS_LD);
    const int trow = lane >> 3, tcol = (lane & 7) * 4;
    // pixel index of row 8q + trow of tile row i (-1: outside the image)
    auto tile_pixel = [&](int i, int q) {
      const int p = (wm * TM + i) * 32 + 8 * q + trow;
      const int oy = ty0 + p / TW, ox = tx0 + p % TW;
      if constexpr ((ABL & 16) != 0) return p;
      return (oy < g.Ho && ox < g.Wo) ? oy * g.Wo + ox : -1;
    };
    // one 32-pixel x 32-channel accumulator tile -> the wave's slab [pixel][LDS_LD floats], scaled
    // (RAW: epilogues that fold the weight scale — a power of two — into their first addition take the accumulators as they are)
    constexpr bool RAW = epi_rawacc<Epi>::value && !epi_flowhead<Epi>::value;
    auto slab_write = [&](const SfAcc& a) __attribute__((always_inline)) {
      const float sc = RAW ? 1.0f : g.wscale;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          *reinterpret_cast<float4*>(tb + (16 * hh + (lane & 15)) * LDS_LD + 16 * cb + 4 * (lane >> 4)) =
              RAW ? make_float4(a.b[hh][cb][0], a.b[hh][cb][1], a.b[hh][cb][2], a.b[hh][cb][3])
                  : make_float4(a.b[hh][cb][0] * sc, a.b[hh][cb][1] * sc, a.b[hh][cb][2] * sc, a.b[hh][cb][3] * sc);
    };
    if constexpr (epi_flowhead<Epi>::value) {
      // ---- flow head: relu(conv1) x conv2's weights, reduced to 18 partial sums per pixel (epilogues_sf.h), then the block adds
      // its 8 waves in fixed order. (Rounds 2-4 did the products on the vector pipe — 36 packed FMAs and 54 DPP adds per group of 8
      // pixels, ~1,600 vector instructions per wave — and the epilogue was a quarter of the kernel: 2.56 -> 2.28 ms per forward.)
      static_assert(!epi_flowhead<Epi>::value || (WM == 1 && TN == 1 && TW == 16), "one block holds all output channels of its pixels");
      constexpr int RP = 20;                                    // floats per pixel row of the exchange (16-byte aligned rows)
      static_assert(32 * RP <= 32 * LDS_LD, "the exchange rows live inside the wave's slab");
      // conv2's 18 partial sums on the matrix engine: [18 -> 32 rows of conv2's weights] x [the wave's 32 channels] x [32 pixels]
      // = two row blocks x two pixel blocks of v_mfma_f32_16x16x32_f16 (K = the wave's 32 channels), three products each.
      // Lane (n, g) of the pixel operand holds relu(conv1 + bias) of pixel 16 pb + n, channels 8 g .. 8 g + 7, read from the
      // transpose slab as two 16-byte pieces; the weight operand (row u = 16 rbk + n, the same 8 channels) lives in registers.
      {
        const int n16 = lane & 15, g16 = lane >> 4;
        const int c8 = n0 + wn * 32 + 8 * g16;
        const float4 bq0 = *reinterpret_cast<const float4*>(ep.bias + c8), bq1 = *reinterpret_cast<const float4*>(ep.bias + c8 + 4);
        const float b8[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
        f16x8 w2h[2], w2l[2];
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
          const int u = 16 * rbk + n16;
          const float* wr = ep.w2 + (long)min(u, 17) * g.N + c8;
          const float4 wa = *reinterpret_cast<const float4*>(wr), wb = *reinterpret_cast<const float4*>(wr + 4);
          const float w8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float w = u < 18 ? w8[e] * ep.w2mul : 0.f;
            w2h[rbk][e] = (_Float16)w;
            w2l[rbk][e] = (_Float16)(w - (float)w2h[rbk][e]);
          }
        }
        bool sat = false;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          slab_write(acc[i][0]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          f16x8 xh[2], xl[2];
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const float* src = tb + (16 * pb + n16) * LDS_LD + 8 * g16;
            const float4 xa = *reinterpret_cast<const float4*>(src), xb = *reinterpret_cast<const float4*>(src + 4);
            const float x8[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const SfPair sp = sf_split_flag(fmaxf(x8[e] + b8[e], 0.f), sat);
              xh[pb][e] = sp.hi; xl[pb][e] = sp.lo;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();   // the slab is consumed: its rows now carry the partial sums
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            f32x4v d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!FAST) {
              d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[0], xl[pb], d0, 0, 0, 0);
              d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[0], xh[pb], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[1], xl[pb], d1, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[1], xh[pb], d1, 0, 0, 0);
            }
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[0], xh[pb], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[1], xh[pb], d1, 0, 0, 0);
            // lane (n, g): sums u = 4 g + e (d0) and 16 + 4 g + e (d1: only u = 16, 17 exist) of pixel 16 pb + n
            float* rp = tb + (16 * pb + n16) * RP;
            *reinterpret_cast<float4*>(rp + 4 * g16) = make_float4(d0[0] * ep.w2inv, d0[1] * ep.w2inv, d0[2] * ep.w2inv, d0[3] * ep.w2inv);
            if (g16 == 0) *reinterpret_cast<float2*>(rp + 16) = make_float2(d1[0] * ep.w2inv, d1[1] * ep.w2inv);
          }
          __syncthreads();   // every wave's partial sums of row tile i are in LDS
          for (int idx = tid; idx < 32 * 18; idx += NT) {
            const int px = idx / 18, u = idx - px * 18;
            const float* rq = reinterpret_cast<const float*>(Pbytes) + px * RP + u;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += rq[w * (32 * LDS_LD)];   // fixed order: deterministic
            const int p = i * 32 + px;
            const int oy = ty0 + p / TW, ox = tx0 + p % TW;
            if (oy < g.Ho && ox < g.Wo)
              ep.G[(long)(n0 / (NW * 32)) * ep.gstride + ((long)img * ep.npix + (long)oy * g.Wo + ox) * 18 + u] = sum;
          }
          __syncthreads();   // the slabs are rewritten by the next row tile
        }
        sf_report(sat);
        return;
      }
    }
    // Epilogues with operand loads (GRU gates, residual adds) are software-pipelined over the wave's tiles: the
    // operands of tile t + 1 are requested BEFORE the stores of tile t are issued. vmcnt retires in order, so a load
    // issued after a store can only be waited for together with that store; this way the wait before tile t + 1's
    // arithmetic counts the younger stores and loads and never drains them.
    struct NoAux {};
    using AuxT = typename std::conditional<Epi::kPrefetch, typename EpiAux4<Epi>::type, NoAux>::type;
    // (only while the operands of a tile fit 32 registers per lane: the q gate's three operand vectors would take the
    // 128-wide block from three to two resident blocks per CU)
    // (round 5 measured the q gate WITH the look-ahead again — its stamps show 11.4 k cycles of operand wait per block: no change,
    // 1.97 / 2.02 ms either way, profiles/r05_ab_q_gate_prefetch.txt: the partner wave's main loop covers the wait)
    constexpr bool PIPE = Epi::kPrefetch && sizeof(AuxT) <= 2 * sizeof(float4);
    AuxT aux_next[PIPE ? 4 : 1];
    bool clamped = false;   // saturation of the sf stores: flagged in a register, reported once (sf.h: sf_store4_flag)
    if constexpr (PIPE) {
      const int nb0 = n0 + (wn * TN) * 32 + tcol;
#pragma unroll
      for (int q = 0; q < 4; ++q) aux_next[q] = ep.load4(img, max(tile_pixel(0, q), 0), min(nb0, g.N - 4));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nb = n0 + (wn * TN + j) * 32 + tcol;
        const float4 bj = ep.bias4(max(min(nb, g.N - 4), 0));   // this lane's 4 channels: one load per channel run
#ifdef ATDN_CONV_STAMP
        unsigned long long st_a = 0;
        if constexpr (kStamp >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_a = __builtin_amdgcn_s_memtime(); }
#endif
        slab_write(acc[i][j]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float4 v[4];
        int mq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = *reinterpret_cast<const float4*>(tb + (8 * q + trow) * LDS_LD + tcol);
          mq[q] = tile_pixel(i, q);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next tile
#ifdef ATDN_CONV_STAMP
        unsigned long long st_b = 0;
        if constexpr (kStamp >= 0) { st_b = __builtin_amdgcn_s_memtime(); st_e_slab += st_b - st_a; }
#endif
        if constexpr (Epi::kPrefetch) {
          AuxT aux[4];
          if constexpr (PIPE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) aux[q] = aux_next[q];
            // next tile of this wave: (i, j + 1) or (i + 1, 0)
            const int jn = (j + 1 < TN) ? j + 1 : 0, in = (j + 1 < TN) ? i : i + 1;
            if (in < TM) {
              const int nbn = n0 + (wn * TN + jn) * 32 + tcol;
#pragma unroll
              for (int q = 0; q < 4; ++q) aux_next[q] = ep.load4(img, max(tile_pixel(in, q), 0), min(nbn, g.N - 4));
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) aux[q] = ep.load4(img, max(mq[q], 0), min(nb, g.N - 4));
          }
#ifdef ATDN_CONV_STAMP
#if ATDN_CONV_STAMP == 2   // variant 2: drain the memory queue before the arithmetic: what the operands (and older stores) still cost here
          unsigned long long st_c = 0;
          if constexpr (kStamp >= 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            st_c = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            st_e_wait += st_c - st_b; st_b = st_c;
          }
#endif
#endif
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (mq[q] >= 0 && nb < g.N) {
              if constexpr (RAW) ep.apply4(img, mq[q], nb, v[q], aux[q], bj, clamped, g.wscale);
              else ep.apply4(img, mq[q], nb, v[q], aux[q], bj, clamped);
            }
#ifdef ATDN_CONV_STAMP
          if constexpr (kStamp >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_e_apply += __builtin_amdgcn_s_memtime() - st_b; }
#endif
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (mq[q] < 0) continue;
            if (nb + 4 <= g.N) {
              if constexpr (RAW) ep.store4((ABL & 16) ? 0 : img, mq[q], nb, v[q], bj, clamped, g.wscale);
              else ep.store4((ABL & 16) ? 0 : img, mq[q], nb, v[q], bj, clamped);
            } else {  // N % 4 != 0: the last run is partial, element-wise (scaled accumulators)
              const float sc = RAW ? g.wscale : 1.0f;
              const float e4[4] = {v[q].x * sc, v[q].y * sc, v[q].z * sc, v[q].w * sc};
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (nb + e < g.N) ep(img, mq[q], nb + e, e4[e]);
            }
          }
        }
      }
    }
    sf_report(clamped);
#ifdef ATDN_CONV_STAMP
    stamp_out();
#endif
    return;
  }
  // ---- pixel-major epilogue (TM x TN tiles per wave). A lane owns NSET = 2 channel columns of a 32 x 32 tile (channel
  // 16 cb + (lane & 15)) with NPX = 8 pixels each: 16 half + 4 (lane >> 4) + k.
  constexpr int NSET = 2, NPX = 8;
  auto col_of = [&](int cs) { return 16 * cs + (lane & 15); };
  auto pix_of = [&](int e) { return 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3); };
  auto val_of = [&](const SfAcc& a, int cs, int e) __attribute__((always_inline)) { return a.b[e >> 2][cs][e & 3]; };
  // per-column constants once per wave, before any store (a load issued after a store waits for that store too)
  typename EpiCol<Epi>::type colj[TN][NSET];
  float biasj[TN][NSET];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int cs = 0; cs < NSET; ++cs) {
      const int n = min(n0 + (wn * TN + j) * 32 + col_of(cs), g.N - 1);
      biasj[j][cs] = 0.f;
      if constexpr (Epi::kStats) biasj[j][cs] = ep.bias[n];
      if constexpr (epi_bias_arg<Epi>::value) colj[j][cs] = ep.col(n);
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pbase = (wm * TM + i) * 32;
    int mm[NPX];
#pragma unroll
    for (int e = 0; e < NPX; ++e) {
      const int p = pbase + pix_of(e);
      const int oy = ty0 + p / TW, ox = tx0 + p % TW;
      mm[e] = (oy < g.Ho && ox < g.Wo) ? oy * g.Wo + ox : -1;
    }
    // statistics epilogues (round 5: instruction count — the kernels with this epilogue ran 3.7-5.2 vector instructions per
    // MFMA): what depends on the pixels alone is formed once per row tile, not once per channel column — the valid-pixel count of
    // the 32-pixel group (the same in every lane: lanes differ in their channel), its reciprocal, the element offsets of the
    // eight pixels; a group without padding pixels (all but the tiles on the right and bottom edge) takes a path without masks;
    // and the stored value is the one the statistics were formed from, not a second bias addition.
    int cnt = 0;
    unsigned mo[NPX];
    if constexpr (Epi::kStats) {
#pragma unroll
      for (int e = 0; e < NPX; ++e) {
        cnt += mm[e] >= 0 ? 1 : 0;
        mo[e] = (unsigned)max(mm[e], 0) * (unsigned)ep.ld;
      }
      cnt += __shfl_xor(cnt, 16);
      cnt += __shfl_xor(cnt, 32);
      cnt = __builtin_amdgcn_readfirstlane(cnt);
    }
    const bool full = cnt == 32;
    const float inv_cnt = 1.0f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int cs = 0; cs < NSET; ++cs) {
        const int n = n0 + (wn * TN + j) * 32 + col_of(cs);
        const bool nok = n < g.N;
        if constexpr (Epi::kStats) {
          // partial statistics of the tile's 32 pixels for this channel: lane-local over its NPX pixels, then across the
          // lanes that hold the same channel (lane ^ 16, lane ^ 32)
          const float bias = nok ? biasj[j][cs] : 0.f;
          float v[NPX];
#pragma unroll
          for (int e = 0; e < NPX; ++e) v[e] = __builtin_fmaf(val_of(acc[i][j], cs, e), g.wscale, bias);
          float sum, m2;
          if (full) {
            sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
          } else {
            sum = 0.f;
#pragma unroll
            for (int e = 0; e < NPX; ++e) sum += mm[e] >= 0 ? v[e] : 0.f;
          }
          sum += __shfl_xor(sum, 16);
          sum += __shfl_xor(sum, 32);
          const float mean = sum * inv_cnt;
          if (full) {
            float d[NPX];
#pragma unroll
            for (int e = 0; e < NPX; ++e) d[e] = v[e] - mean;
            m2 = __builtin_fmaf(d[1], d[1], d[0] * d[0]);
#pragma unroll
            for (int e = 2; e < NPX; ++e) m2 = __builtin_fmaf(d[e], d[e], m2);
          } else {
            m2 = 0.f;
#pragma unroll
            for (int e = 0; e < NPX; ++e) { const float dd = mm[e] >= 0 ? v[e] - mean : 0.f; m2 = __builtin_fmaf(dd, dd, m2); }
          }
          m2 += __shfl_xor(m2, 16);
          m2 += __shfl_xor(m2, 32);
          const int grp = tloc * (TH * TW / 32) + wm * TM + i;
          if ((lane >> 4) == 0 && nok) {
            const long o = ((long)img * ep.groups_per_img + grp) * g.N + n;
            ep.part_sum[o] = sum;
            ep.part_m2[o] = m2;
          }
          if (lane == 0 && n == 0) ep.part_cnt[(long)img * ep.groups_per_img + grp] = (float)cnt;
          if (nok) {   // the raw value: scalar base of the image's slice + 32-bit element offset
            char* ob = reinterpret_cast<char*>(ep.dst + (long)img * ep.ob);
#pragma unroll
            for (int e = 0; e < NPX; ++e)
              if (full || mm[e] >= 0) *reinterpret_cast<float*>(ob + 4u * (mo[e] + (unsigned)n)) = v[e];
          }
          continue;
        }
        if (nok) {
          if constexpr (Epi::kPrefetch) {
            typename Epi::Aux aux[NPX];
#pragma unroll
            for (int e = 0; e < NPX; ++e) aux[e] = ep.load(img, max(mm[e], 0), n);
#pragma unroll
            for (int e = 0; e < NPX; ++e)
              if (mm[e] >= 0) ep.apply(img, mm[e], n, val_of(acc[i][j], cs, e) * g.wscale, aux[e]);
          } else if constexpr (epi_bias_arg<Epi>::value) {
#pragma unroll
            for (int e = 0; e < NPX; ++e)
              if (mm[e] >= 0) ep.store_c(img, mm[e], n, val_of(acc[i][j], cs, e) * g.wscale, colj[j][cs]);
          } else {
#pragma unroll
            for (int e = 0; e < NPX; ++e)
              if (mm[e] >= 0) ep(img, mm[e], n, val_of(acc[i][j], cs, e) * g.wscale);
          }
        }
      }
  }
}

template <int TH, int TW, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST = false, bool NORM = false, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64, 1) void conv_sf6_kernel(const Conv2Geom g, const Epi ep) {
  conv_sf6_body<TH, TW, BN, WM, WN, KH, KW, Epi, FAST, NORM, ABL>(g, ep, (int)blockIdx.x);
}

// Two INDEPENDENT convolutions of the same instantiation as one launch (round 5: convc2 and convf2 of the motion encoder — the
// correlation branch and the flow branch of update.py:76-92 do not depend on each other). Blocks [0, nblk0) are the first
// convolution's, the rest the second's; the second starts at a multiple of 8 so that both keep their XCD mapping (blocks of the
// gap exit at once). A block computes exactly what it computes in a launch of its own: the results are the same bits; what changes
// is that the second convolution's blocks fill the first one's tail instead of waiting for it.
__device__ __forceinline__ Conv2Geom conv2_geom_pick(const Conv2Geom& a, const Conv2Geom& b, bool s) {
  // field by field (uniform selects: the geometry stays in scalar registers; an indexed array of kernel arguments is spilled to LDS)
  Conv2Geom g;
  g.src0 = s ? b.src0 : a.src0; g.src1 = s ? b.src1 : a.src1; g.sb0 = s ? b.sb0 : a.sb0; g.sb1 = s ? b.sb1 : a.sb1;
  g.ld0 = s ? b.ld0 : a.ld0; g.ld1 = s ? b.ld1 : a.ld1; g.C0 = s ? b.C0 : a.C0; g.C1 = s ? b.C1 : a.C1;
  g.H = s ? b.H : a.H; g.W = s ? b.W : a.W; g.Ho = s ? b.Ho : a.Ho; g.Wo = s ? b.Wo : a.Wo;
  g.KH = s ? b.KH : a.KH; g.KW = s ? b.KW : a.KW; g.padH = s ? b.padH : a.padH; g.padW = s ? b.padW : a.padW;
  g.PH = s ? b.PH : a.PH; g.PW = s ? b.PW : a.PW; g.tiles_x = s ? b.tiles_x : a.tiles_x; g.tiles_y = s ? b.tiles_y : a.tiles_y;
  g.nimg = s ? b.nimg : a.nimg; g.ntile_n = s ? b.ntile_n : a.ntile_n;
  g.w = s ? b.w : a.w; g.ldw = s ? b.ldw : a.ldw; g.N = s ? b.N : a.N; g.wscale = s ? b.wscale : a.wscale;
  g.in_mean = nullptr; g.in_rstd = nullptr;
  return g;
}
template <int TH, int TW, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST = false>
__global__ __launch_bounds__(WM * WN * 64, 1) void conv_sf6_pair_kernel(const Conv2Geom g0, const Conv2Geom g1, const Epi ep0,
                                                                        const Epi ep1, const int nblk0, const int start1) {
  const int b = (int)blockIdx.x;
  if (b >= nblk0 && b < start1) return;
  const bool second = b >= start1;
  const Conv2Geom g = conv2_geom_pick(g0, g1, second);
  const Epi ep = Epi::pick(ep0, ep1, second);
  // (one call: the body's LDS image exists once)
  conv_sf6_body<TH, TW, BN, WM, WN, KH, KW, Epi, FAST, false, 0>(g, ep, second ? b - start1 : b);
}

template <int TH, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST = false, bool NORM = false, int ABL = 0>
inline Conv2Geom conv_sf6_geom(const ConvShape& s, float wscale) {
  constexpr int TW = 16;
  ATDN_CHECK(s.KH == KH && s.KW == KW && s.stride == 1, "kernel shape does not match the instantiation");
  Conv2Geom g{};
  g.src0 = s.src0; g.src1 = s.src1; g.sb0 = s.sb0; g.sb1 = s.sb1; g.ld0 = s.ld0; g.ld1 = s.ld1;
  g.C0 = s.C0; g.C1 = s.C1; g.H = s.H; g.W = s.W;
  g.KH = s.KH; g.KW = s.KW; g.padH = s.padH; g.padW = s.padW;
  g.Ho = conv_out(s.H, s.KH, 1, s.padH); g.Wo = conv_out(s.W, s.KW, 1, s.padW);
  g.PH = TH + s.KH - 1; g.PW = TW + s.KW - 1;
  ATDN_CHECK(s.C0 % 32 == 0 && s.C1 % 32 == 0 && s.C0 > 0 && s.ld0 % 4 == 0, "TAP-mode channel constraints");
  ATDN_CHECK(s.ldw % 4 == 0 && s.ldw >= s.KH * s.KW * (s.C0 + s.C1), "weight rows too short");
  ATDN_CHECK(!epi_vec4<Epi>::value || !Epi::kPrefetch || s.N % 4 == 0, "channel-vector epilogue with operand loads needs N % 4 == 0");
  ATDN_CHECK((long)s.H * s.W < (1L << 20) - 1, "image too large for the packed patch descriptor");
  // the epilogues address an image's slice as a scalar base + an unsigned 32-bit BYTE offset (sf.h: sf_store4_flag_u / sf_load4u,
  // the statistics epilogue's mo[e]): pixels x row width x 4 B must stay under 4 GB. Rows of this library are <= 1024 floats.
  ATDN_CHECK((long)g.Ho * g.Wo * 1024L * 4L <= (1L << 32) && s.ld0 <= 1024 && s.ld1 <= 1024 && s.N <= 1024,
             "per-image slice of 4 GB or more: the 32-bit epilogue offsets would wrap");
  g.tiles_x = cdiv(g.Wo, TW); g.tiles_y = cdiv(g.Ho, TH);
  g.nimg = s.nimg; g.ntile_n = cdiv(s.N, BN);
  g.w = s.wfrag16; g.ldw = s.ldw; g.N = s.N; g.wscale = wscale;
  g.in_mean = s.in_mean; g.in_rstd = s.in_rstd;
  ATDN_CHECK(NORM == (s.in_mean != nullptr) && (!NORM || (s.in_rstd && s.C1 == 0 && s.ld0 == s.C0)),
             "normalise-on-load: one dense fp32 source with its mean / rstd");
  ATDN_CHECK(g.w != nullptr, "missing weight copy for this kernel");
  return g;
}

template <int TH, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST = false, bool NORM = false, int ABL = 0>
inline void launch_conv_sf6(const ConvShape& s, float wscale, Epi ep, hipStream_t st) {
  constexpr int TW = 16;
  const Conv2Geom g = conv_sf6_geom<TH, BN, WM, WN, KH, KW, Epi, FAST, NORM, ABL>(s, wscale);
  set_groups(ep, g.tiles_x * g.tiles_y * (TH * TW / 32));
  const int nblk = g.nimg * g.tiles_x * g.tiles_y * g.ntile_n;
  hipLaunchKernelGGL((conv_sf6_kernel<TH, TW, BN, WM, WN, KH, KW, Epi, FAST, NORM, ABL>), dim3(nblk), dim3(WM * WN * 64), 0, st, g, ep);
  ATDN_HIP(hipGetLastError());
}

template <int TH, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST>
inline void launch_conv_sf6_pair(const ConvShape& s0, float wscale0, Epi ep0, const ConvShape& s1, float wscale1, Epi ep1,
                                 hipStream_t st) {
  constexpr int TW = 16;
  static_assert(!Epi::kStats, "the pair launch serves plain-store epilogues");
  const Conv2Geom g0 = conv_sf6_geom<TH, BN, WM, WN, KH, KW, Epi, FAST, false, 0>(s0, wscale0);
  const Conv2Geom g1 = conv_sf6_geom<TH, BN, WM, WN, KH, KW, Epi, FAST, false, 0>(s1, wscale1);
  const int nblk0 = g0.nimg * g0.tiles_x * g0.tiles_y * g0.ntile_n;
  const int start1 = (nblk0 + 7) / 8 * 8;
  const int nblk1 = g1.nimg * g1.tiles_x * g1.tiles_y * g1.ntile_n;
  hipLaunchKernelGGL((conv_sf6_pair_kernel<TH, TW, BN, WM, WN, KH, KW, Epi, FAST>), dim3(start1 + nblk1), dim3(WM * WN * 64), 0, st,
                     g0, g1, ep0, ep1, nblk0, start1);
  ATDN_HIP(hipGetLastError());
}


// block width the 3x3 dispatch below picks for a layer (the flow-head fusion needs one block to hold all channels)
inline int conv_sf6_block_width_3x3(const ConvShape& s) {
  const int Ho = conv_out(s.H, s.KH, 1, s.padH), Wo = conv_out(s.W, s.KW, 1, s.padW);
  const long tiles = (long)s.nimg * cdiv(Wo, 16) * cdiv(Ho, 8);
  int bn = 64;
  if (s.N <= 32) bn = 32;
  else if (s.N == 96) bn = 96;
  else {
    int best = cdiv(s.N, 64) * 64;
    for (int c : {128, 256})
      if (cdiv(s.N, c) * c <= best) { best = cdiv(s.N, c) * c; bn = c; }
  }
  while (bn > 64 && bn != 96 && tiles * cdiv(s.N, bn) < 300) bn /= 2;
  return bn;
}

template <int TH, int BN, int WM, int WN, int KH, int KW, class Epi, bool FAST, bool NORM>
inline void launch_conv_sf6_m(const ConvShape& s, float wscale, const Epi& ep, hipStream_t st) {
  launch_conv_sf6<TH, BN, WM, WN, KH, KW, Epi, FAST, NORM>(s, wscale, ep, st);
}

// Picks the block shape for N output channels and launches the fragment-major-weight kernel: 8x16-pixel tiles, or
// 12x16 for the 64- and 96-wide blocks (3 MFMA row tiles per wave: less halo, fewer tile seams; measured
// 7-10 % faster on the encoder shapes and on N = 192) when the taller tiles pad the image no worse and still
// cover the chip. Returns false when this path does not serve the shape (the caller falls back to the plain implicit GEMM).
template <int KH, int KW, class Epi, bool FAST>
inline bool conv_sf6_try_shape(const ConvShape& s, float wscale, const Epi& ep, hipStream_t st, int* bn_out, int* th_out) {
  const int Ho = conv_out(s.H, s.KH, 1, s.padH), Wo = conv_out(s.W, s.KW, 1, s.padW);
  const long tiles = (long)s.nimg * cdiv(Wo, 16) * cdiv(Ho, 8);
  // block width: the one of {256, 128, 64} that pads N least (ties: the widest, it shares the patch among more
  // channels); N = 96 has its own 2x3-wave block. Measured (tools/microbench_conv.py, B = 8): N = 256 -> 256-wide
  // 163 us vs 64-wide 174 us; N = 192 -> 64-wide 154 us vs 128- or 256-wide 186 us.
  int bn = 64;
  if (s.N <= 32 && KH == 3) bn = 32;       // flow head (N = 2): 4 waves of 32 px x 32 ch, half the padding of a 64-wide block
  else if (s.N == 96 && KH == 3) bn = 96;
  else {
    int best = cdiv(s.N, 64) * 64;
    for (int c : {128, 256})
      if (cdiv(s.N, c) * c <= best) { best = cdiv(s.N, c) * c; bn = c; }
  }
  // small grids: narrower blocks (more of them) until the chip is covered
  while (bn > 64 && bn != 96 && tiles * cdiv(s.N, bn) < 300) bn /= 2;
  // ConvGRU convolutions (1x5 / 5x1): 128-wide blocks even for N = 256. Their epilogue is heavy (gate operands, sigmoids,
  // two stores per value: ~20 % of a block's time) and no MFMA overlaps it inside a block; three 4-wave blocks fit a CU
  // where one 8-wave block does, so one block's epilogue runs under the others' main loops (z|r, 16 pairs: 295 -> 286 us).
  // (Round 5 also measured 64-wide blocks for the gates, 8 x 16 and 12 x 16 pixels, 2 x 2 waves — the shape the 3x3 convolutions
  // of the motion encoder run best with: z|r 3.40 -> 3.52 / 3.63 ms per forward, q 1.94 -> 2.06 / 2.08; the vertical pass worse
  // still, 174 registers = two waves per SIMD without the wider block's patch reuse. profiles/r05_ab_gru_block_width.txt)
  if constexpr (KH != 3) bn = std::min(bn, 128);
  *th_out = 8;
  if (s.in_mean) {   // normalise-on-load: statistics convs of the feature network (3x3, 64 / 96 / 128 channels)
    if constexpr (Epi::kStats && KH == 3 && !FAST) {
      // (statistics convolutions: the tile height fixes which 32 pixels form a statistics group, so it must follow from
      // the layer's geometry alone, never from the number of images in the launch — a clip, a continued clip and single
      // pairs then produce bit-identical InstanceNorm statistics)
      const bool tall = (bn == 64 || bn == 96) && cdiv(Ho, 12) * 12 * 100 <= cdiv(Ho, 8) * 8 * 103;
      *bn_out = bn; *th_out = tall ? 12 : 8;
      if (bn == 64 && tall) launch_conv_sf6_m<12, 64, 2, 2, KH, KW, Epi, false, true>(s, wscale, ep, st);
      else if (bn == 64) launch_conv_sf6_m<8, 64, 2, 2, KH, KW, Epi, false, true>(s, wscale, ep, st);
      else if (bn == 96 && tall) launch_conv_sf6_m<12, 96, 2, 3, KH, KW, Epi, false, true>(s, wscale, ep, st);
      else if (bn == 96) launch_conv_sf6_m<8, 96, 2, 3, KH, KW, Epi, false, true>(s, wscale, ep, st);
      else if (bn == 128) launch_conv_sf6_m<8, 128, 1, 4, KH, KW, Epi, false, true>(s, wscale, ep, st);
      else return false;
      return true;
    } else {
      return false;
    }
  }
  // Small grids (round 6: one or two pairs per launch — the reference's per-frame call pattern; 47 x 154 pixels are 60 tiles of
  // 8 x 16): when even 64-wide blocks leave most of the chip's 512 block slots empty, 4 x 16-pixel tiles double the blocks. A
  // block then carries half the MFMAs behind a relatively larger halo, which costs efficiency nobody is short of at this size.
  // Same K order as every other tile shape, so the same bits (tests/test_gpu_parity.py: a pair comes out of an 8-pair launch
  // exactly as out of a single-pair call). Not for the statistics epilogues (their tile height follows from the layer's geometry
  // alone) nor the fused flow head (one block holds all channels of its pixels).
  if constexpr (!Epi::kStats && !epi_flowhead<Epi>::value) {
    // (ATDN_CONV_SMALL_TILES=1: the 4 x 16 x 64 form for every 1x5 / 5x1 convolution at ANY grid size — the A/B of DESIGN.md 10.8)
    static const bool force_small = getenv("ATDN_CONV_SMALL_TILES") && getenv("ATDN_CONV_SMALL_TILES")[0] == '1';
    if ((bn == 64 && tiles * cdiv(s.N, 64) < 300) || (force_small && KH != 3)) {
      *bn_out = 64; *th_out = 4;
      launch_conv_sf6_m<4, 64, 2, 2, KH, KW, Epi, FAST, false>(s, wscale, ep, st);
      return true;
    }
  }
  if constexpr (KH != 3 && !Epi::kStats && !epi_flowhead<Epi>::value) {
    // (ATDN_CONV_SMALL_TILES=2: 4 x 16-pixel x 128-channel blocks for the 1x5 / 5x1 convolutions — half the row tiles per wave of the
    // shipped 8 x 16 x 128 block, so fewer registers: the A/B of DESIGN.md 10.8)
    static const bool half128 = getenv("ATDN_CONV_SMALL_TILES") && getenv("ATDN_CONV_SMALL_TILES")[0] == '2';
    if (half128 && bn == 128) {
      *bn_out = 128; *th_out = 4;
      launch_conv_sf6_m<4, 128, 1, 4, KH, KW, Epi, FAST, false>(s, wscale, ep, st);
      return true;
    }
  }
  if constexpr (KH == 3) {
    if (bn == 32) { *bn_out = 32; launch_conv_sf6_m<8, 32, 4, 1, KH, KW, Epi, FAST, false>(s, wscale, ep, st); return true; }
    const long tiles12 = (long)s.nimg * cdiv(Wo, 16) * cdiv(Ho, 12);
    const bool tall = (bn == 64 || bn == 96) && cdiv(Ho, 12) * 12 * 100 <= cdiv(Ho, 8) * 8 * 103 &&
                      (Epi::kStats || tiles12 * cdiv(s.N, bn) >= 512);   // (statistics: geometry only, see above)
    if (tall) {
      *bn_out = bn; *th_out = 12;
      if (bn == 64) launch_conv_sf6_m<12, 64, 2, 2, KH, KW, Epi, FAST, false>(s, wscale, ep, st);
      else launch_conv_sf6_m<12, 96, 2, 3, KH, KW, Epi, FAST, false>(s, wscale, ep, st);
      return true;
    }
  }
  *bn_out = bn;
  switch (bn) {
    case 64:  launch_conv_sf6_m<8, 64, 2, 2, KH, KW, Epi, FAST, false>(s, wscale, ep, st); return true;  // 4 waves of 64 px x 32 ch
    case 128: launch_conv_sf6_m<8, 128, 1, 4, KH, KW, Epi, FAST, false>(s, wscale, ep, st); return true;
    case 256: launch_conv_sf6_m<8, 256, 1, 8, KH, KW, Epi, FAST, false>(s, wscale, ep, st); return true;
    default: break;
  }
  if constexpr (KH == 3) {
    if (bn == 96) { launch_conv_sf6_m<8, 96, 2, 3, KH, KW, Epi, FAST, false>(s, wscale, ep, st); return true; }
  }
  return false;
}

// Both convolutions as ONE launch when each of them, alone, would run the 12 x 16-pixel x 64-channel 3x3 kernel with a plain
// store (the same selection rules as conv_sf6_try_shape<3, 3>: the pair changes the schedule, never the kernel a convolution gets).
template <class Epi>
inline bool conv_sf6_try_pair(const ConvShape& s0, float wscale0, const Epi& ep0, const ConvShape& s1, float wscale1, const Epi& ep1,
                              hipStream_t st, bool fast) {
  if constexpr ((epi_gen6<Epi>::value & 1) == 0 || Epi::kStats || epi_flowhead<Epi>::value) {
    return false;
  } else {
    auto picks_tall64 = [&](const ConvShape& s) {
      if (!conv_halo_eligible(s) || s.KH != 3 || s.KW != 3 || !s.wfrag16 || s.in_mean) return false;
      if (epi_vec4<Epi>::value && (s.N % 4) != 0) return false;
      if (s.C0 % 32 != 0 || s.C1 % 32 != 0 || s.C0 <= 0 || s.ld0 % 4 != 0) return false;
      if (conv_sf6_block_width_3x3(s) != 64) return false;
      const int Ho = conv_out(s.H, 3, 1, s.padH), Wo = conv_out(s.W, 3, 1, s.padW);
      const long tiles12 = (long)s.nimg * cdiv(Wo, 16) * cdiv(Ho, 12);
      return cdiv(Ho, 12) * 12 * 100 <= cdiv(Ho, 8) * 8 * 103 && tiles12 * cdiv(s.N, 64) >= 512;
    };
    if (!picks_tall64(s0) || !picks_tall64(s1)) return false;
    if (fast) launch_conv_sf6_pair<12, 64, 2, 2, 3, 3, Epi, true>(s0, wscale0, ep0, s1, wscale1, ep1, st);
    else launch_conv_sf6_pair<12, 64, 2, 2, 3, 3, Epi, false>(s0, wscale0, ep0, s1, wscale1, ep1, st);
    return true;
  }
}

template <class Epi>
inline bool conv_sf6_try(const ConvShape& s, float wscale, const Epi& ep, hipStream_t st, int* bn_out, int* th_out, bool fast) {
  constexpr int kinds = epi_gen6<Epi>::value;
  if (kinds == 0 || !s.wfrag16 || s.stride != 1) return false;
  if (epi_vec4<Epi>::value && Epi::kPrefetch && (s.N % 4) != 0) return false;
  if (epi_vec4<Epi>::value && s.N < 4) return false;
  if (s.C0 % 32 != 0 || s.C1 % 32 != 0 || s.C0 <= 0 || s.ld0 % 4 != 0) return false;
  if constexpr ((kinds & 1) != 0) {
    if (s.KH == 3 && s.KW == 3)
      return fast ? conv_sf6_try_shape<3, 3, Epi, true>(s, wscale, ep, st, bn_out, th_out)
                  : conv_sf6_try_shape<3, 3, Epi, false>(s, wscale, ep, st, bn_out, th_out);
  }
  if constexpr ((kinds & 2) != 0) {
    if (s.KH == 1 && s.KW == 5)
      return fast ? conv_sf6_try_shape<1, 5, Epi, true>(s, wscale, ep, st, bn_out, th_out)
                  : conv_sf6_try_shape<1, 5, Epi, false>(s, wscale, ep, st, bn_out, th_out);
    if (s.KH == 5 && s.KW == 1)
      return fast ? conv_sf6_try_shape<5, 1, Epi, true>(s, wscale, ep, st, bn_out, th_out)
                  : conv_sf6_try_shape<5, 1, Epi, false>(s, wscale, ep, st, bn_out, th_out);
  }
  return false;
}

}  // namespace atdn
