// GMA attention on the split-f16 engine (see attention.h for the data layout).
//
//   qk_softmax_kernel<STATS>   Q K^T with the row softmax fused in (gma.py:60-74). A block owns 128 query rows (one
//       32-row strip per wave, its 32 x 128 query fragments resident in registers) and sweeps ALL keys in 64-key tiles
//       staged through LDS. Operands are swapped (keys = MFMA rows), so a lane owns ONE query row: running maximum
//       and running sum need no cross-lane traffic. STATS = true is the cheap first pass (hi x hi products only ->
//       rowmax~); STATS = false recomputes the logits at full split-f16 precision, writes e = exp(s - rowmax~) in
//       fragment-major order (1 KiB per wave store, non-temporal) and the row sums. The fp32 logits (1.68 GB at 8 pairs)
//       never exist in memory; the separate softmax pass (read 1.68 GB, write 1.68 GB) is gone.
//   attn_v3_kernel             attention x V (gma.py:102-115) as a two-set ping-pong, described at the kernel. Every wave
//       streams ITS strip of the attention matrix — one contiguous 0.5-0.7 MB run — straight into MFMA operand registers,
//       1 KiB per wave load; only V^T (L2-resident, shared by the block) goes through LDS. (The 4-wave kernel it replaced
//       — every wave interleaving its own memory work with its own MFMAs — and the 4-byte SF4 element format were
//       deleted in round 3 after their A/B tests; measurements in DESIGN.md 3.6.)
#include "attention.h"
#include <cstdlib>

#include <type_traits>

#include "conv_mfma.h"
#include "sf.h"

namespace atdn {
namespace {

constexpr float LOG2E = 1.4426950408889634f;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 ld_frag(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ f16x8 ld_frag_nt(const char* p) {
  const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return __builtin_bit_cast(f16x8, t);
}
__device__ __forceinline__ void st_frag_nt(char* p, f16x8 v) {
  __builtin_nontemporal_store(__builtin_bit_cast(v4f, v), reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// ---- H3 element format (attention.h): hi = f16(e); residual byte in units of 2^(E - 33), E = f16 exponent (>= 1) of the
// largest hi of the group of eight
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// (values past the f16 range — a full-precision logit more than ~4.16 above the first pass's f16 x f16 row maximum — are
// clamped to 65504 and flagged: the caller reports them through the sf saturation counter, sf_report)
// Round 4: the encode is the vector-instruction bulk of the softmax kernel (6.4 vector instructions per MFMA on the SQ counters),
// so it is written for instruction count: `gsum` = the sum of the group's eight values (the caller needs it for the row sum
// anyway) bounds every one of them and carries a NaN, so ONE test per group replaces eight and the clamp moves to a cold path;
// the group maximum is two v_max3 levels; the residual byte is v_cvt_pk_u8_f32(fma(v - hi, 2^(33 - E), 128)) — that conversion
// rounds to nearest even and saturates to [0, 255] (tools/diag/cvt_u8_check.hip), which is what rint, min(., 127) and "+ 128"
// spelt out before (a tie rounds the same way; a quotient of +128 still stores 127).
__device__ __forceinline__ void h3_encode(const float* v_in, float gsum, f16x8& hi, u32x2& bytes, bool& clamped) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = v_in[i];
  if (__builtin_expect(!(gsum <= 65504.f), 0)) {   // rare: some value may be past the f16 range (or a NaN)
    asm volatile("; h3_encode: saturated group");   // (keeps this a branch: if-converted, its 8 tests and 8 clamps run for every group)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      clamped |= !(v[i] <= 65504.f);
      v[i] = fminf(v[i], 65504.f);
    }
  }
  float hf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (_Float16)v[i];
    hf[i] = (float)hi[i];
  }
  const float mx = fmaxf(fmaxf(fmaxf(hf[0], hf[1]), fmaxf(hf[2], hf[3])), fmaxf(fmaxf(hf[4], hf[5]), fmaxf(hf[6], hf[7])));
  // exponent field of the float = f16 exponent + 112; f16 subnormals (and zero) count as E = 1
  const unsigned ef = max(__float_as_uint(mx) >> 23, 113u);
  const float inv_unit = __uint_as_float((272u - ef) << 23);   // 2^(33 - E)
  unsigned w0 = 0, w1 = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // |v - hi| <= ulp(hi) / 2 <= ulp(group) / 2: the quotient lies in [-128, 128]
    const float q = __builtin_fmaf(v[i] - hf[i], inv_unit, 128.f);
    if (i < 4) w0 = __builtin_amdgcn_cvt_pk_u8_f32(q, i, w0); else w1 = __builtin_amdgcn_cvt_pk_u8_f32(q, i - 4, w1);
  }
  bytes[0] = w0;
  bytes[1] = w1;
}
__device__ __forceinline__ f16x8 h3_decode_lo(f16x8 hi, u32x2 bytes) {
  // Round 4: the decode runs in a memory phase BESIDE the partner wave's MFMAs, and every vector instruction of it takes
  // ~4 cycles of the SIMD's issue away from them (s_memtime stamps, DESIGN.md 8.7: a multiply phase took 1280-1350 cycles for
  // 768 cycles of MFMAs). So the instruction count is what matters here:
  //   * group maximum on the BIT PATTERNS (values are non-negative: f16 order = unsigned order) with integer maxima instead
  //     of 7 compare + select pairs;
  //   * residual byte -> float with v_cvt_f32_ubyteN, and the "- 128" folded into one FMA: (q - 128) * unit =
  //     fma(q, unit, -128 unit), exact (|q - 128| <= 128 times a power of two) — written any other way the compiler
  //     subtracts in the integer domain first (v_lshrrev + v_add_u32 + v_cvt_f32_i32 + v_mul).
  // (A packed-f16 form — byte permute to 1024 + byte, v_pk_add_f16, v_pk_mul_f16 — was measured 27 % SLOWER in round 2:
  // packed VALU beside MFMAs is an anti-lever on this chip, MI355X_MICROARCH.md.)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 hb = __builtin_bit_cast(u32x4, hi);
  // sign bits are clear, so an unsigned 32-bit maximum picks the dword with the largest UPPER half; the lower halves separately
  const unsigned up = max(max(hb[0], hb[1]), max(hb[2], hb[3]));
  const unsigned lw = max(max(hb[0] & 0xFFFFu, hb[1] & 0xFFFFu), max(hb[2] & 0xFFFFu, hb[3] & 0xFFFFu));
  const unsigned e = max(max(up >> 26, lw >> 10), 1u);
  const float unit = __uint_as_float((e + 94u) << 23);        // 2^(E - 33)
  const float off = -128.f * unit;
  f16x8 lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned w = bytes[i >> 2];
    const float q = (float)((w >> (8 * (i & 3))) & 255u);
    float r = __builtin_fmaf(q, unit, off);   // exact while representable
    asm("" : "+v"(r));                         // opaque: keeps the SLP vectoriser from pairing the FMAs into v_pk_fma_f32 (below)
    lo[i] = (_Float16)r;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------------ Q K^T + softmax
template <bool STATS, bool FAST>
__global__ __launch_bounds__(256, 2) void qk_softmax_kernel(const float* __restrict__ qk, const AttnGeom g,
                                                           const float* __restrict__ rowmax_in,
                                                           float* __restrict__ rowmax_out, float* __restrict__ P,
                                                           float* __restrict__ rinv) {
  // Round 3: on v_mfma_f32_16x16x32_f16 (K = 32 = one channel chunk per instruction). Keys are the row operand (four blocks
  // of 16 keys per 64-key tile), the wave's strip of 32 queries the column operand (two blocks rb of 16): lane (n = lane & 15,
  // g = lane >> 4) holds, for query 32 strip + 16 rb + n, the keys 16 kb + 4 g + 0..3 of key block kb. A 32-key chunk is two
  // key blocks, so the lane's eight values of a chunk are the keys 4 g + (i & 3) + 16 (i >> 2) — exactly one entry
  // (rb, lane) of the stored block (attention.h): the store is one contiguous KiB per wave instruction again.
  constexpr int KROW = 160;            // LDS pitch of a key row: conflict-free ds_read_b128 for the 16x16x32 lane map
  constexpr int KT = 64;               // keys per LDS tile
  constexpr int CH = KT * KROW;        // one 32-channel chunk of the tile
  constexpr int IMG = 4 * CH;          // [4 channel chunks][64 keys][160 B]
  constexpr bool FULL = !STATS && !FAST;   // all three products of the split
  // Row blocks of 16 queries per wave. The statistics sweep (one f16 product per logit, nothing stored) takes FOUR: its loop is
  // bound by the LDS reads of the key fragments — every wave reads the whole key tile — and 64 queries per wave halve them per
  // MFMA (round 5: 0.333 -> 0.262 ms per forward, same row maxima bit for bit: profiles/r05_ab_rowmax_64.txt). The full sweep stores its 32-query strips in the consumer's block order: two.
  constexpr int RB = STATS ? 4 : 2;
  constexpr int SR = 16 * RB;              // queries per wave (strip)
  constexpr int BLK = AT_BLK_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[2 * IMG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nstrip = STATS ? (g.Npad + SR - 1) / SR : g.RT;
  const int tiles = (nstrip + 3) >> 2;
  const int id = xcd_remap(blockIdx.x, g.B * tiles);
  const int b = id / tiles, tile = id - b * tiles;
  const int strip = tile * 4 + wave;
  const bool strip_ok = strip < nstrip;
  const int n16 = lane & 15, g16 = lane >> 4;
  const char* qkb = reinterpret_cast<const char*>(qk + (long)b * g.N * 256);

  // query fragments (the column operand), resident for the whole sweep: [row block][channel chunk]
  f16x8 qh[RB][4], ql[RB][4];
  int mrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    mrow[rb] = strip * SR + 16 * rb + n16;
    const char* qrow = qkb + (long)min(mrow[rb], g.N - 1) * 1024 + 16 * g16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qh[rb][c] = ld_frag(qrow + c * 128);
      if (FULL) ql[rb][c] = ld_frag(qrow + c * 128 + 64);
    }
  }

  // key tile loader: thread -> key rows lr, lr + 32 of the tile, 16-byte slot ls of every 128-byte channel chunk
  const int lr = tid >> 3, ls = tid & 7;
  v4f kreg[2][4];
  auto fetch = [&](int j) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = min(j * KT + lr + 32 * i, g.N - 1);   // keys past N: clamped, their columns are masked below
      const char* src = qkb + (long)n * 1024 + 512 + 16 * ls;
#pragma unroll
      for (int c = 0; c < 4; ++c) kreg[i][c] = *reinterpret_cast<const v4f*>(src + c * 128);
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<v4f*>(lds + buf * IMG + c * CH + (lr + 32 * i) * KROW + 16 * ls) = kreg[i][c];
  };

  const int NHT = (g.Q + 1) >> 1;
  float run[RB];         // running row maximum / running row sum of this lane's keys, per row block
  float c0[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    c0[rb] = 0.f;
    run[rb] = STATS ? -INFINITY : 0.f;
    if (!STATS) c0[rb] = (float)AT_SHIFT - rowmax_in[(long)b * g.Npad + min(mrow[rb], g.Npad - 1)] * LOG2E;
  }
  bool clamped = false;                  // saturation of the f16 store, reported once after the sweep
  char* pdst = reinterpret_cast<char*>(P) + ((long)(b * g.RT + min(strip, g.RT - 1)) * g.Q) * BLK;

  fetch(0);
  stash(0);
  __syncthreads();
  fetch(min(1, NHT - 1));
  for (int j = 0; j < NHT; ++j) {
    const char* img = lds + (j & 1) * IMG + n16 * KROW + 16 * g16;
    f32x4v acc[4][RB];  // [key block][row block]
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[kb][rb][e] = 0.f;
    // key fragments of channel chunk c + 1 are requested before the MFMAs of chunk c are issued (two register sets);
    // left to itself the compiler reads each fragment right before its MFMA and waits out the LDS round trip
    f16x8 kh[2][4], kl[2][4];   // [set][key block]
    auto read_keys = [&](int set, int c) __attribute__((always_inline)) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const char* kp = img + c * CH + 16 * kb * KROW;
        kh[set][kb] = ld_frag(kp);
        if (FULL) kl[set][kb] = ld_frag(kp + 64);
      }
    };
    read_keys(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c + 1 < 4) read_keys((c + 1) & 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          f32x4v a = acc[kb][rb];
          if (FULL) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[c & 1][kb], qh[rb][c], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[c & 1][kb], ql[rb][c], a, 0, 0, 0);
          }
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[c & 1][kb], qh[rb][c], a, 0, 0, 0);
          acc[kb][rb] = a;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next tile goes into the other image: it was last read in iteration j - 1 and every wave has passed the
    // barrier since; the tile after that is requested now and lands during the epilogue and the next MFMA block
    stash((j + 1) & 1);
    fetch(min(j + 2, NHT - 1));
    // Only the LAST tile of a sweep can hold keys past N: every other tile takes the body without the per-value key test (a
    // compare and a select per value in the kernel's vector-instruction-bound epilogue). Rows past N (the padding rows of the
    // last strip: copies of row N - 1, never read back — rinv is 0 for them and the consumer drops them) are not masked at all.
    auto epilogue = [&](auto tail_tag) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
      for (int cq = 0; cq < 2; ++cq) {        // the two 32-key chunks of the tile
        const int key0 = j * KT + 32 * cq + 4 * g16;
        u32x2 bytes2[2] = {{0u, 0u}, {0u, 0u}};   // residual bytes of both row blocks: one 16-byte store (and one load in the consumer)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if constexpr (STATS) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (!TAIL || key0 + (i & 3) + 16 * (i >> 2) < g.N) run[rb] = fmaxf(run[rb], acc[2 * cq + (i >> 2)][rb][i & 3]);
          } else {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float x = __builtin_amdgcn_exp2f(fmaf(acc[2 * cq + (i >> 2)][rb][i & 3], LOG2E, c0[rb]));
              v[i] = (!TAIL || key0 + (i & 3) + 16 * (i >> 2) < g.N) ? x : 0.f;
            }
            // the group's sum: its share of the row sum, and the bound h3_encode tests the group's range with
            const float gsum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            run[rb] += gsum;
            const int q = 2 * j + cq;
            if (strip_ok && q < g.Q) {
              char* d = pdst + (long)q * BLK;
              f16x8 hi;
              u32x2 bytes;
              h3_encode(v, gsum, hi, bytes, clamped);
              st_frag_nt(d + rb * 1024 + lane * 16, hi);
              bytes2[rb] = bytes;
            }
          }
        }
        if (!STATS && strip_ok && 2 * j + cq < g.Q) {
          typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
          const u32x4s bb = {bytes2[0][0], bytes2[0][1], bytes2[1][0], bytes2[1][1]};
          __builtin_nontemporal_store(bb, reinterpret_cast<u32x4s*>(pdst + (long)(2 * j + cq) * BLK + 2048 + lane * 16));
        }
      }
    };
    if ((j + 1) * KT > g.N) {
      asm volatile("; qk_softmax: tail tile");   // (a real branch: if-converted, the tail's tests would run for every tile)
      epilogue(std::true_type{});
    } else {
      epilogue(std::false_type{});
    }
    __syncthreads();
  }
  sf_report(clamped);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float r_ = run[rb];
    if (STATS) {
      r_ = fmaxf(r_, __shfl_xor(r_, 16));
      r_ = fmaxf(r_, __shfl_xor(r_, 32));
      if (g16 == 0 && strip_ok && mrow[rb] < g.Npad) rowmax_out[(long)b * g.Npad + mrow[rb]] = r_;
    } else {
      r_ += __shfl_xor(r_, 16);
      r_ += __shfl_xor(r_, 32);
      if (g16 == 0 && strip_ok) rinv[(long)b * g.Npad + mrow[rb]] = (mrow[rb] < g.N && r_ > 0.f) ? 1.0f / r_ : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention x V
// ---- attention x V as a two-set ping-pong (MI355X_MICROARCH.md, "Two waves per SIMD"). A block is EIGHT waves (one strip
// each, 256 registers: ONE block per CU — 16 pairs are two rounds of blocks): waves 0-3 (set A) and 4-7 (set B) share the
// four SIMDs pairwise and alternate roles every phase, separated by a barrier:
//     phase 2q    : A multiplies chunk q (48 MFMAs, every operand already in registers) and, in the gaps between them,
//                     decodes the H3 residuals of ITS chunk q + 1
//                   B does memory work: attention loads two chunks ahead, V^T fragments of chunk q from LDS into registers
//     phase 2q + 1: B multiplies chunk q and decodes its chunk q + 1
//                   A does memory work: stages V^T chunk q + 2 into the LDS ring (three images), attention loads, fragments
//                     for chunk q + 1
// Round 4 put s_memtime stamps around the segments of a phase (tools/diag/attn_stamps.py, profiles/r04_attn_stamps*.txt).
// Before: a multiply phase took 1280-1350 cycles for 768 cycles of MFMAs, and it was what the barrier waited for. Two causes:
// (1) the ~100 vector instructions of the partner's decode — the two waves of a SIMD share its vector issue, and an MFMA
// itself holds it for 8 of its 16 cycles; (2) for set B the compiler had SUNK the whole decode behind the barrier into the
// `if (q < Q)` block, in front of the wave's own MFMAs. Now the decode is 62 instructions (h3_decode_lo) and rides in the MFMA
// gaps of the wave that owns it (one or two per gap: 'vector-instruction ISSUE cost' of the guide says that is nearly free),
// and the memory phases carry loads, LDS traffic and barriers only: 632 -> 526 us per launch at 16 pairs (same job).
// History: the 4-wave kernel this replaced (every wave interleaving its own memory work with its own MFMAs) ran 372 us at 8
// pairs against 315 for the first ping-pong. Measured in round 4 and not kept (profiles/r04_ab_attention.txt; priority 1 and 2 again in round 5 with the
// decode in the MFMA gaps, after it had paid in the conv kernels: 6.85 / 6.49 ms per forward for the product against 6.83 / 6.50
// and 6.57 / 6.65 — inside the run-to-run spread of this stage, profiles/r05_ab_attn_prio.txt): s_setprio(1)
// around the multiplying set's MFMAs; the V^T staging split over both sets (it only moves work between two memory phases).
#ifdef ATDN_ATTN_STAMP   // diagnostic: s_memtime stamps around the segments of a phase, summed per wave (waves 0 and 4 of every block)
__device__ unsigned atdn_attn_stamps_dev[1024][2][8];
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    seg[k] += (unsigned)(t_ - tlast); tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
// SPLIT (round 6: the low-latency form for the per-frame callers, B <= 4): the key axis of a (pair, 8 strips) tile is cut into
// `nsplit` ranges of `qsplit` chunks, one block each — ONE pair is 29 tiles on 256 CUs otherwise (attention x V was 2.4 of the
// 7.1 ms of a single-pair forward) — and a block stores its raw fp32 sums into its slab of `part` [nsplit][B][Npad][128];
// attn_reduce_kernel adds the slabs in order and applies the residual. Same kernel body, another summation order: the
// default path (SPLIT = false) keeps one order for every batch size, which is what makes clip mode bit-identical to pair mode.
template <bool FAST, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void attn_v3_kernel(const float* __restrict__ P, const float* __restrict__ rinv,
                                                        const AttnGeom g, const float* __restrict__ vT,
                                                        const float* __restrict__ gamma, const float* __restrict__ mf,
                                                        float* __restrict__ out, const long sb, const int ld,
                                                        const int qsplit = 0, float* __restrict__ part = nullptr) {
  // Round 3: the multiply runs on v_mfma_f32_16x16x32_f16 (K = 32 = one whole chunk of keys per instruction; same FLOP per
  // cycle, but the chip holds a higher clock under this shape on random operands: §3.10 of DESIGN.md). A = V^T (16 channels
  // per block), B = the attention fragment (16 query rows per block): lane (n = lane & 15, g = lane >> 4) holds row / column
  // n and the eight keys of group g. 8 channel blocks x 2 row blocks x 3 products = 48 MFMAs per chunk, as many matrix-pipe
  // cycles as the 24 MFMAs of the 32x32x16 form; the V^T image has a 160-byte row pitch (conflict-free for this lane map).
  constexpr int VROW = 160;
  constexpr int IMG = 128 * VROW;      // V^T chunk: [128 channels][160 B]
  constexpr int BLK = AT_BLK_BYTES;
  constexpr int D = 3;                 // attention chunks resident per wave (ring of register sets; 4 measured 1-2 % slower, twice)
  constexpr int DEC2 = 14;             // MFMA gaps that carry two instructions of the decode (the other 34 carry one: 62 in all)
  __shared__ __attribute__((aligned(16))) char lds[3 * IMG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool setA = wave < 4;
  const int tiles = (g.RT + 7) >> 3;
  const int nsplit = SPLIT ? (g.Q + qsplit - 1) / qsplit : 1;
  const int id0 = xcd_remap(blockIdx.x, g.B * tiles * nsplit);
  const int ks = SPLIT ? id0 % nsplit : 0;
  const int id = SPLIT ? id0 / nsplit : id0;
  const int b = id / tiles, tile = id - b * tiles;
  const int strip = tile * 8 + wave;
  const bool strip_ok = strip < g.RT;
  const int n16 = lane & 15, g16 = lane >> 4;
  const int qlo = SPLIT ? ks * qsplit : 0;             // first chunk of this block's key range
  const int Q = SPLIT ? min(qsplit, g.Q - qlo) : g.Q;  // ... and its length: every chunk index below is relative to qlo

  // (address arithmetic of the two streams on the scalar unit — wave index through readfirstlane, uniform bases + 32-bit lane
  // offsets — was measured in round 4: 1-3 % SLOWER on the stage, not kept)
  const char* pblk = reinterpret_cast<const char*>(P) + ((long)(b * g.RT + min(strip, g.RT - 1)) * g.Q + qlo) * BLK;
  typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
  f16x8 ring[D][2];      // [slot][row block]
  u32x4s ringb[D];       // residual bytes of both row blocks: one load
  auto loadP = [&](int q, int slot) __attribute__((always_inline)) {
    const char* p = pblk + (long)min(q, Q - 1) * BLK;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) ring[slot][rb] = ld_frag_nt(p + rb * 1024 + lane * 16);
    ringb[slot] = __builtin_nontemporal_load(reinterpret_cast<const u32x4s*>(p + 2048 + lane * 16));
  };
  auto bytes_of = [&](int slot, int rb) __attribute__((always_inline)) { return u32x2{ringb[slot][2 * rb], ringb[slot][2 * rb + 1]}; };

  // V^T staging (set A only: tid < 256): rows lr + 32 i, 16-byte slot ls of the row's 128-byte [32 hi | 32 lo] chunk. The
  // producer of V^T (epilogues_sf.h: SfVT) already stores the keys of a chunk in operand order — slot g = keys
  // 4 g + (i & 3) + 16 (i >> 2), the order of the attention fragments — so a slot is copied verbatim: one ds_write_b128
  // per row, eight lanes = one contiguous 128-byte row, no bank conflicts (the 8-byte piece moves this replaced had
  // SQ_LDS_BANK_CONFLICT at 17 % of the kernel's LDS cycles).
  const int lr = (tid & 255) >> 3, ls = tid & 7;
  const float* vrow = vT + ((long)b * 128 + lr) * g.ldN + 4 * ls + qlo * 32;
  constexpr int NSTG = 4;
  v4f breg[2][NSTG];
  auto fetchB = [&](int q, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NSTG; ++i) breg[set][i] = *reinterpret_cast<const v4f*>(vrow + (long)(32 * i) * g.ldN + min(q, Q - 1) * 32);
  };
  const int boff = lr * VROW + 16 * ls;
  auto stashB = [&](int image, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NSTG; ++i) *reinterpret_cast<v4f*>(lds + image * IMG + 32 * i * VROW + boff) = breg[set][i];
  };

  f32x4v acc[8][2];      // [channel block][row block]
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cb][rb][e] = 0.f;

  // operands of the chunk about to be multiplied, all in registers
  f16x8 vh[8], vl[8], ph[2], pl[2];
  const int foff = n16 * VROW + 16 * g16;
  auto read_frags = [&](int image) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const char* vp = lds + image * IMG + foff + 16 * cb * VROW;
      vh[cb] = ld_frag(vp);
      if (!FAST) vl[cb] = ld_frag(vp + 64);
    }
  };
  auto take_chunk = [&](int slot) __attribute__((always_inline)) {   // attention operands of a chunk out of the ring
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      ph[rb] = ring[slot][rb];
      if (!FAST) pl[rb] = h3_decode_lo(ph[rb], bytes_of(slot, rb));
    }
    // (pinned here: the results are only used inside `if (q < Q)`, and left alone the compiler sinks the decode into that block)
    if (!FAST) asm volatile("" :: "v"(pl[0]), "v"(pl[1]));
  };
  auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        f32x4v c = acc[cb][rb];
        if (!FAST) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[cb], ph[rb], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[cb], pl[rb], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[cb], ph[rb], c, 0, 0, 0);
        acc[cb][rb] = c;
      }
  };
  // multiply chunk q AND decode the residuals of chunk q + 1 (ring slot `next`) in the gaps between the MFMAs: a 16x16x32 MFMA
  // holds the SIMD's vector issue for 8 of its 16 cycles, so one or two vector instructions of the SAME wave ride in every gap
  // nearly free (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), where the same instructions in the partner's memory
  // phase lengthen that phase — which is what the barrier waits for once the decode is short
  auto multiply_next = [&](int next) __attribute__((always_inline)) {
    f16x8 pln[2];
    if (!FAST) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) pln[rb] = h3_decode_lo(ring[next][rb], bytes_of(next, rb));
    }
    multiply();
    if (!FAST) {
#pragma unroll
      for (int k = 0; k < DEC2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
#pragma unroll
      for (int k = DEC2; k < 48; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      ph[rb] = ring[next][rb];
      if (!FAST) pl[rb] = pln[rb];
    }
  };

  constexpr int U = (D % 2) ? 2 * D : D;        // unroll: ring slots (D) x V^T register sets (2)
#ifdef ATDN_ATTN_STAMP
  unsigned seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = 0;
#endif
  const int nq = (Q + U - 1) / U * U;           // surplus chunks multiply nothing (uniform branch), barriers still match
  if (setA) {
    // prologue: V^T chunks 0, 1 -> images 0, 1; chunks 2, 3 wait in the register sets; attention chunks 0..2
    fetchB(0, 0);
    fetchB(1, 1);
#pragma unroll
    for (int c = 0; c < D; ++c) loadP(c, c);
    stashB(0, 0);
    stashB(1, 1);
    fetchB(2, 0);
    fetchB(3, 1);
    __syncthreads();                            // (1) images 0 and 1 published
    read_frags(0);
    take_chunk(0);
    int image = 0;                              // LDS image of chunk q: q % 3
#ifdef ATDN_ATTN_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tlast = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    for (int q0 = 0; q0 < nq; q0 += U) {
#pragma unroll
      for (int d = 0; d < U; ++d) {
        const int q = q0 + d;
        const int image1 = image == 2 ? 0 : image + 1, image2 = image1 == 2 ? 0 : image1 + 1;
        // phase 2q: multiply chunk q
        __builtin_amdgcn_sched_barrier(0);
        if (q < Q) multiply_next((d + 1) % D);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(0);
        __syncthreads();
        STAMP(1);
        // phase 2q + 1: memory work. V^T chunk q + 2 (requested two iterations ago) -> image (q + 2) % 3, last read in
        // phase 2q - 2; its register set takes chunk q + 4; the ring slot of chunk q takes chunk q + 3
        stashB(image2, d & 1);
        fetchB(q + 4, d & 1);
        loadP(q + D, d % D);
        STAMP(2);
        read_frags(image1);                     // chunk q + 1: staged in phase 2q - 1, published by its barrier
        image = image1;
        STAMP(3);
        __syncthreads();
        STAMP(5);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < D - 1; ++c) loadP(c, c);
    take_chunk(0);
    __syncthreads();                            // (1)
    int image = 0;
#ifdef ATDN_ATTN_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tlast = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    for (int q0 = 0; q0 < nq; q0 += U) {
#pragma unroll
      for (int d = 0; d < U; ++d) {
        const int q = q0 + d;
        // phase 2q: memory work: attention chunk q + 2 into the slot chunk q - 1 left, operands of chunk q
        loadP(q + D - 1, (d + D - 1) % D);
        STAMP(0);
        read_frags(image);
        image = image == 2 ? 0 : image + 1;
        STAMP(1);
        __syncthreads();
        STAMP(3);
        // phase 2q + 1: multiply chunk q
        __builtin_amdgcn_sched_barrier(0);
        if (q < Q) multiply_next((d + 1) % D);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        __syncthreads();
        STAMP(5);
      }
    }
  }

#ifdef ATDN_ATTN_STAMP
  if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 1024) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atdn_attn_stamps_dev[blockIdx.x][wave >> 2][k] = seg[k];
  }
#endif
  // lane (n, g) holds, for query rows 32 strip + 16 rb + n, channels 16 cb + 4 g + 0..3
  if constexpr (SPLIT) {
    float* pp = part + ((long)ks * g.B + b) * g.Npad * 128;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = strip * 32 + 16 * rb + n16;
      if (strip_ok && m < g.N) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
          *reinterpret_cast<float4*>(pp + (long)m * 128 + 16 * cb + 4 * g16) =
              make_float4(acc[cb][rb][0], acc[cb][rb][1], acc[cb][rb][2], acc[cb][rb][3]);
      }
    }
    return;
  }
  const float gam = gamma[0];
  const float* mfb = mf + (long)b * sb;
  float* ob = out + (long)b * sb;
  bool clamped = false;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int m = strip * 32 + 16 * rb + n16;
    if (strip_ok && m < g.N) {
      const float rv = rinv[(long)b * g.Npad + m] * gam;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        const int c = 16 * cb + 4 * g16;
        const float4 x = sf_load4(mfb, (long)m * ld, c);
        const float4 o = make_float4(x.x + rv * acc[cb][rb][0], x.y + rv * acc[cb][rb][1], x.z + rv * acc[cb][rb][2],
                                     x.w + rv * acc[cb][rb][3]);
        sf_store4_flag(ob, (long)m * ld, c, o, clamped);
      }
    }
  }
  sf_report(clamped);
}

// the split form's second half: out[b][m][c] = mf + gamma * rinv[m] * (slab 0 + slab 1 + ... in order), stored as sf
__global__ __launch_bounds__(256) void attn_reduce_kernel(const float* __restrict__ part, const int nsplit, const AttnGeom g,
                                                          const float* __restrict__ rinv, const float* __restrict__ gamma,
                                                          const float* __restrict__ mf, float* __restrict__ out, const long sb,
                                                          const int ld) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one thread = 4 channels of one query row
  const long rows = (long)g.B * g.N;
  bool clamped = false;
  if (i < rows * 32) {
    const long row = i >> 5;
    const int c = (int)(i & 31) * 4;
    const int b = (int)(row / g.N), m = (int)(row - (long)b * g.N);
    const long slab = (long)g.B * g.Npad * 128;
    const float* p = part + ((long)b * g.Npad + m) * 128 + c;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int k = 1; k < nsplit; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(p + k * slab);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const float rv = rinv[(long)b * g.Npad + m] * gamma[0];
    const float4 x = sf_load4(mf + (long)b * sb, (long)m * ld, c);
    sf_store4_flag(out + (long)b * sb, (long)m * ld, c, make_float4(x.x + rv * s.x, x.y + rv * s.y, x.z + rv * s.z, x.w + rv * s.w),
                   clamped);
  }
  sf_report(clamped);
}

// one wave per (pair, strip, chunk) block: normalised probabilities as fp32 rows (tests / debug only)
__global__ __launch_bounds__(64) void attn_decode_kernel(const float* __restrict__ P, const float* __restrict__ rinv,
                                                         const AttnGeom g, float* __restrict__ rows) {
  const long blk = blockIdx.x;
  const int q = (int)(blk % g.Q);
  const long bs = blk / g.Q;
  const int strip = (int)(bs % g.RT), b = (int)(bs / g.RT);
  const int lane = threadIdx.x, n = lane & 15, gq = lane >> 4;
  const char* p = reinterpret_cast<const char*>(P) + blk * AT_BLK_BYTES;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int m = strip * 32 + 16 * rb + n;
    if (m >= g.N) continue;
    const float rv = rinv[(long)b * g.Npad + m];
    float* row = rows + ((long)b * g.N + m) * g.ldN + q * 32;
    const f16x8 hi = ld_frag(p + rb * 1024 + lane * 16);
    const f16x8 lo = h3_decode_lo(hi, *reinterpret_cast<const u32x2*>(p + 2048 + lane * 16 + rb * 8));
    // key group g of the producer: keys 4 g + (i & 3) + 16 (i >> 2)
#pragma unroll
    for (int i = 0; i < 8; ++i) row[4 * gq + (i & 3) + 16 * (i >> 2)] = ((float)hi[i] + (float)lo[i]) * rv;
  }
}

}  // namespace

#ifdef ATDN_ATTN_STAMP
extern "C" int atdn_attn_stamps(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(atdn_attn_stamps_dev), sizeof(unsigned) * 1024 * 2 * 8);
}
#endif

void launch_qk_rowmax(const float* qk, const AttnGeom& g, float* rowmax, hipStream_t st) {
  const int nblk = g.B * (((g.Npad + 63) / 64 + 3) / 4);   // 64-query strips, four per block
  hipLaunchKernelGGL((qk_softmax_kernel<true, false>), dim3(nblk), dim3(256), 0, st, qk, g, nullptr, rowmax, nullptr, nullptr);
  ATDN_HIP(hipGetLastError());
}

void launch_qk_softmax(const float* qk, const AttnGeom& g, const float* rowmax, float* P, float* rinv, bool fast,
                       hipStream_t st) {
  const int nblk = g.B * ((g.RT + 3) / 4);
  const dim3 gr(nblk), bl(256);
  if (fast) hipLaunchKernelGGL((qk_softmax_kernel<false, true>), gr, bl, 0, st, qk, g, rowmax, nullptr, P, rinv);
  else hipLaunchKernelGGL((qk_softmax_kernel<false, false>), gr, bl, 0, st, qk, g, rowmax, nullptr, P, rinv);
  ATDN_HIP(hipGetLastError());
}

int attn_v_splits(const AttnGeom& g) {
  // enough key ranges to put a block on (nearly) every CU, at most 8: one pair = 29 tiles -> 8 x 29 chunks
  const int tiles = g.B * ((g.RT + 7) / 8);
  int n = std::min(8, std::max(1, 256 / std::max(tiles, 1)));
  while (n > 1 && (g.Q + n - 1) / n < 8) --n;   // a range of fewer than 8 chunks is all prologue
  return n;
}

void launch_attn_v(const float* P, const float* rinv, const AttnGeom& g, const float* vT, const float* gamma,
                   const float* mf, float* out, long sb, int ld, bool fast, hipStream_t st, float* part) {
  ATDN_CHECK(g.ldN % 32 == 0 && g.Q * 32 == g.ldN && ld % 32 == 0, "attention geometry");
  // (ATDN_ATTN_FORCE_SPLIT=n: the experiment of DESIGN.md section 10.4 — the split form at ANY batch size, n key ranges)
  static const int force = getenv("ATDN_ATTN_FORCE_SPLIT") ? atoi(getenv("ATDN_ATTN_FORCE_SPLIT")) : 0;
  const int want = !part ? 1 : force > 0 ? std::min(force, 8) : attn_v_splits(g);
  if (want > 1) {
    const int qsplit = (g.Q + want - 1) / want, nsplit = (g.Q + qsplit - 1) / qsplit;
    const dim3 gr(g.B * ((g.RT + 7) / 8) * nsplit), bl(512);
    if (fast) hipLaunchKernelGGL((attn_v3_kernel<true, true>), gr, bl, 0, st, P, rinv, g, vT, gamma, mf, out, sb, ld, qsplit, part);
    else hipLaunchKernelGGL((attn_v3_kernel<false, true>), gr, bl, 0, st, P, rinv, g, vT, gamma, mf, out, sb, ld, qsplit, part);
    ATDN_HIP(hipGetLastError());
    const long n4 = (long)g.B * g.N * 32;
    hipLaunchKernelGGL(attn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, nsplit, g, rinv, gamma, mf, out, sb, ld);
    ATDN_HIP(hipGetLastError());
    return;
  }
  const dim3 gr(g.B * ((g.RT + 7) / 8)), bl(512);
  if (fast) hipLaunchKernelGGL((attn_v3_kernel<true>), gr, bl, 0, st, P, rinv, g, vT, gamma, mf, out, sb, ld);
  else hipLaunchKernelGGL((attn_v3_kernel<false>), gr, bl, 0, st, P, rinv, g, vT, gamma, mf, out, sb, ld);
  ATDN_HIP(hipGetLastError());
}

void launch_attn_decode(const float* P, const float* rinv, const AttnGeom& g, float* rows, hipStream_t st) {
  const long nblk = (long)g.B * g.RT * g.Q;
  hipLaunchKernelGGL(attn_decode_kernel, dim3((unsigned)nblk), dim3(64), 0, st, P, rinv, g, rows);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
