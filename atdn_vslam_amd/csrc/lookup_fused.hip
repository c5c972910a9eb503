// Correlation-pyramid lookup on a BRICKED pyramid, fused with the 1x1 convolution behind it (split-f16 pipeline).
//
// Reference: CorrBlock.__call__ + bilinear_sampler (whl:GMA/core/corr.py:32-53, utils/utils.py:59-73) followed by
// BasicMotionEncoder.convc1 (whl:GMA/core/update.py:76-78): corr = relu(convc1(lookup(coords))).
//
// Round 1 kept level l of the pyramid row-major, [source pixel][H_l * W_l] floats, and gathered a 12 x 12 window per
// (pixel, level) with scalar loads: twelve 48-byte row segments from rows 616 bytes apart, 3.7x the bytes the lookup
// needs on the HBM side (PMC), then wrote the 324 samples (352-channel sf rows, 81.5 MB per iteration at 8 pairs) for
// the 1x1 convolution to read back. Here
//   * a level is stored in BRICKS of 4 rows x 8 columns = 32 floats = one 128-byte line, brick-major inside PIXEL BLOCKS of 64
//     consecutive source pixels of a pair (round 4; kernels.h: BrickPyramid) — a block of this kernel is one pixel block:
//       level[(((pair * NPB + (p >> 6)) * NBK_l + by * BW_l + bx) * 64 + (p & 63)) * 32 + (y & 3) * 8 + (x & 7)], by = y >> 2, bx = x >> 3,
//     cells past H_l / W_l are zeros (they ARE grid_sample's zero padding for taps just outside). The correlation kernel
//     (corr_bricks.hip) writes this layout by itself: its "target pixel" operand is a copy of the feature map in brick order
//     with zero rows for the padding cells, so output column n' is the brick address. Levels 1-3 come from 2x2-pooled
//     features (correlation is linear in the target features) the same way. Neighbouring source pixels, whose windows
//     mostly share bricks, read neighbouring lines.
//   * a 12 x 12 window touches at most 3 x 4 bricks: one wave fetches them as whole lines (8 lanes x 16 B per brick, two
//     load instructions), three (pixel, level) units ahead;
//   * a block owns 64 source pixels (round 3; 32 before): its sixteen waves (level = wave & 3, sixteen pixels each) sample into
//     a 64 x 352 split-f16 tile in LDS — the A operand of the 1x1 convolution — and then multiply it with the fragment-major weights (straight
//     from L2 into operand registers, as the halo kernels do) and store relu(. + bias) as sf rows. The sampled
//     correlation features never touch HBM.
// Sample arithmetic is that of lookup_sf_kernel (kernels.hip), step by step, so results are unchanged.
#include "conv_mfma.h"
#include "kernels.h"
#include "sf.h"

namespace atdn {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LP = kBrickPixelBlock;   // pixels per pixel block of the bricked pyramid (the layout's unit, kernels.h)
constexpr int TP = 32;            // source pixels per block of THIS kernel: half a pixel block. Round 3 went from 32 to 64 (one block of 16
                                  // waves per CU: the convc1 phase was bound by the L1 rate of its weight loads, and a block multiplies
                                  // twice the pixels with every weight fragment it fetches). Round 5, with the sampling phase on its
                                  // diet, that phase was 35 % of the kernel and strictly serial behind the sampling (ONE block per CU:
                                  // nothing overlaps anything, not even the next block's start-up): back to 32 pixels and 8 waves so
                                  // that TWO blocks share a CU and one's matrix phase runs under the other's sampling: 183 -> 171 us
                                  // per launch (profiles/r05_ab_lookup_half_blocks.txt).
constexpr int NCH = 352;          // 4 * 81 samples padded to a multiple of 32
constexpr int APITCH = 1504;      // bytes per pixel row of the A tile: 11 x 128 + 96 = 94 slots of 16 B, 94 = 2 (mod 4): conflict-free
                                  // ds_read_b128 for the operand lane map of v_mfma_f32_16x16x32_f16 (16 pixels x 4 slots per instruction).
                                  // The 96 bytes behind the 11 chunks are never read: bytes 1408-1439 and 1472-1503 are the DUMP of the
                                  // lanes without a sample (their hi / lo stores, 64 bytes apart like everybody's)
constexpr int ADUMP = 1408;
constexpr int GW = 24, GH = 24;   // per-wave window grid: 3 x 4 bricks (24 x 16 cells). Pitch 24 dwords = -8 (mod 32 banks): the four
                                  // rows x two 16-byte halves a brick's eight lanes write (ds_write_b128: groups of 8 lanes) are eight
                                  // distinct bank quads, and a 32-lane group of sample reads (ds_read_b32) laid out as 8 x-positions x 4
                                  // rows is 32 distinct banks. (Pitch 28 with samples in channel order: SQ_LDS_BANK_CONFLICT = 25 % of
                                  // the kernel's LDS cycles.) Rows 16-23 are a dump for the lanes without a brick: the unit
                                  // body has NO branch (a branch around an LDS store made the compiler drain every prefetched load,
                                  // vmcnt(0), per unit)
constexpr int DEPTH = 4;          // (pixel, level) units in flight per wave (a multiple of the four levels; 8 in flight: 4 % SLOWER, round 5)
constexpr int NWAVE = TP / 4;     // waves per block: wave w samples all four levels of pixels 4 w .. 4 w + 3
constexpr int UPW = TP * 4 / NWAVE;   // units (pixels of its level) per wave

// One (pixel, level) unit — round 5: the sampling unit on an instruction diet (rounds 3-4: ~101 vector ALU + 15 LDS instructions
// per unit in the ISA, the kernel's bound; now ~35 + 10).
//  * ONE x chain and ONE y chain per unit (the reference's arithmetic, corr.py:43-49 and utils.py:63-70, evaluated for the window's
//    centre, offset 0) instead of nine + nine: all 81 samples of a unit share one integer origin and — up to the last bits of the
//    reference's normalise / denormalise round trip, which differ from offset to offset by a few ulp of the coordinate — one
//    fractional part. The four bilinear products are therefore ONE set of numbers per unit, a lane's grid address is a per-lane
//    constant plus a scalar, and the window is 10 x 10 cells (origin = floor - 4), not 12 x 12: fewer bricks fetched.
//  * Those per-unit numbers are computed ONCE PER WAVE, lane u = unit u (16 units: 4 pixels x 4 levels), before the loop — not
//    by all 64 lanes redundantly, unit after unit — and reach the unit that needs them as scalars (v_readlane): the origin for
//    the scalar-unit arithmetic of the brick range, the four products as scalar operands of the sample FMAs.
//  * Which of a unit's 3 x 4 bricks exist (inside the window AND inside the map) is a 12-bit scalar mask, blown up to a lane
//    mask by s_bitreplicate; a lane's load offset is a per-level lane constant plus a scalar and ONE select.
//  * saturation: the largest sample magnitude is tracked (one v_max3 per unit) and compared once, after the loop.
//  * the residual of the split, lo = f16(v - hi), is one v_fma_mixlo_f16 (bit-identical to convert-subtract-convert:
//    tools/diag/sf_mix_check.hip).
// Rounds 3-4 evaluated the 18 chains on lanes 0-17 into an LDS table (one ds_write_b64, four ds_read_b64 and two fences per
// unit) and formed the four products in every lane, twice.

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// FUSED: multiply the sampled tile with convc1 and store relu(. + bias) as sf rows [pixel][256];
// !FUSED: store the sampled tile itself as sf rows [pixel][352] (debug reads, the unfused comparison path)
template <bool FUSED, bool FAST>
__global__ __launch_bounds__(NWAVE * 64, TP == 32 ? 4 : 1) void lookup_conv_kernel(const BrickPyramid pyr, const float* __restrict__ coords1,
                                                            const long npix, float* __restrict__ coords_used,
                                                            const float* __restrict__ wfrag, const float wscale,
                                                            const float* __restrict__ bias, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char atile[TP * APITCH];
  __shared__ __attribute__((aligned(16))) float grid[NWAVE][GH * GW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (tells the compiler what the hardware guarantees: wave-uniform)
  // block = TP consecutive pixels of one pair: a pixel block of the pyramid, or half of one (the last block of a pair is partial)
  static_assert(LP % TP == 0 && TP % 16 == 0 && TP <= 64, "a block of this kernel is a whole fraction of a pixel block");
  const int bpp = (pyr.N + TP - 1) / TP;                        // blocks per pair
  const int pair = (int)blockIdx.x / bpp, kblk = (int)blockIdx.x - pair * bpp;
  const int pblk = (kblk * TP) / LP, poff = (kblk * TP) % LP;   // the pyramid's pixel block, and this block's first line inside its brick runs
  const long p0 = (long)pair * pyr.N + (long)kblk * TP;         // first pixel of the block in coords1 / out
  const int np = min(TP, pyr.N - kblk * TP);
  (void)npix;

  // pad channels 324..351 of every row are zero
  for (int i = tid; i < TP * 28; i += NWAVE * 64) {
    const int row = i / 28, c = 324 + (i - row * 28);
    _Float16* q = reinterpret_cast<_Float16*>(atile + row * APITCH + (c >> 5) * 128) + (c & 31);
    q[0] = (_Float16)0.f;
    q[32] = (_Float16)0.f;
  }

  // ---- phase 1: wave w samples ALL FOUR levels of the pixels 4 w .. 4 w + 3. (Rounds 2 and early 3 gave a wave one level of
  // sixteen pixels: the four level-0 waves — whose bricks come from HBM, 222 MB of volume per pair, while levels 1-3 mostly
  // hit L2 / the Infinity Cache — set the pace and the other twelve waited at the block barrier: `SQ_WAIT_ANY` 0.45.) Units run
  // in the order (pixel, level) with the level fastest, so unit pi has level pi & 3 = its slot in the DEPTH-4 ring: every
  // level-dependent quantity is indexed by a compile-time slot.
  static_assert(DEPTH % 4 == 0, "unit pi has level pi & 3 = (pi % DEPTH) & 3");
  const int pbase = wave * (UPW / 4);
  // coordinates of the block's pixels: lane i holds pixel i (pixels past the end repeat the last one; never stored)
  const long pc = p0 + min(lane, np - 1);
  const float2 cmine = *reinterpret_cast<const float2*>(coords1 + pc * 2);
  if (wave == 0 && lane < np && coords_used) *reinterpret_cast<float2*>(coords_used + (p0 + lane) * 2) = cmine;

  // ---- the wave's 16 units at once: lane u (and u + 16, u + 32, u + 48: copies) = unit u = (pixel pbase + (u >> 2), level u & 3)
  // per unit: P_m12 = which of its 3 x 4 bricks exist (bit 3 r + c), P_sb = byte offset of its window's first brick in the pixel
  // block's level (without the pixel's line), P_gb = byte offset of its window's first cell in the wave's grid, and the four
  // bilinear products
  int P_m12, P_sb, P_gb;
  float P_w00, P_w01, P_w10, P_w11;
  {
#pragma clang fp contract(off)   // the reference's roundings, one operation at a time
    const int myu = lane & 15, lev = myu & 3;
    const int mypp = min(pbase + (myu >> 2), np - 1);
    const float cx = __shfl(cmine.x, mypp), cy = __shfl(cmine.y, mypp);
    const int Wl = lev == 0 ? pyr.W[0] : lev == 1 ? pyr.W[1] : lev == 2 ? pyr.W[2] : pyr.W[3];
    const int Hl = lev == 0 ? pyr.H[0] : lev == 1 ? pyr.H[1] : lev == 2 ? pyr.H[2] : pyr.H[3];
    const float inv = lev == 0 ? 1.0f : lev == 1 ? 0.5f : lev == 2 ? 0.25f : 0.125f;     // coords / 2**l (corr.py:44)
    float xc = cx * inv, yc = cy * inv;
    const bool sane = (fabsf(xc) < 1.0e6f) && (fabsf(yc) < 1.0e6f);   // also rejects NaN
    // a unit that is not sane samples zeros (every brick of its window is outside the map): finite coordinates keep NaN / inf
    // out of the weights, and its origin goes to -2^24
    xc = sane ? xc : 0.f; yc = sane ? yc : 0.f;
    // pos = c + d; g = 2 pos / (S - 1) - 1; u = (g + 1) * ((S - 1) / 2): the reference's round trip through normalised
    // coordinates, for d = 0. The division is a multiplication by the correctly rounded reciprocal plus one FMA correction
    // step. That is the correctly rounded quotient whenever the first product is within one ulp of it (Markstein) — not a
    // theorem for every (t, b); for the divisors this network has (W_l - 1, H_l - 1 of the four levels at C1 and C2)
    // tests/test_host_arith.py checks the chain against the IEEE division over every position on a 1/64-pixel lattice across
    // the maps plus 16 pixels of margin.
    auto chain = [](float c0, float sz) __attribute__((always_inline)) {
      const float rs = 1.0f / sz, hs = sz / 2.f;                      // (IEEE divisions, once per wave)
      const float t = 2.f * c0;
      float q = t * rs;
      q = __builtin_fmaf(__builtin_fmaf(-q, sz, t), rs, q);          // = t / sz (see above)
      return ((q - 1.f) + 1.f) * hs;
    };
    const float ux = chain(xc, (float)(Wl - 1)), uy = chain(yc, (float)(Hl - 1));
    const float flx = floorf(ux), fly = floorf(uy);
    const float ww = ux - flx, nn = uy - fly, ee = 1.f - ww, ss = 1.f - nn;
    P_w00 = ee * ss; P_w01 = ww * ss; P_w10 = ee * nn; P_w11 = ww * nn;
    // window = the 10 x 10 cells floor - 4 .. floor + 5 (offsets -4 .. 4, taps at +0 and +1)
    const int wx0 = sane ? (int)flx - 4 : -(1 << 24), wy0 = sane ? (int)fly - 4 : -(1 << 24);
    const int bx0 = wx0 >> 3, by0 = wy0 >> 2;                       // arithmetic shifts: floor for negative origins
    P_gb = sane ? ((wy0 - 4 * by0) * GW + (wx0 - 8 * bx0)) * 4 : 0;
    const int BWl = lev == 0 ? pyr.BW[0] : lev == 1 ? pyr.BW[1] : lev == 2 ? pyr.BW[2] : pyr.BW[3];
    const int BHl = lev == 0 ? pyr.BH[0] : lev == 1 ? pyr.BH[1] : lev == 2 ? pyr.BH[2] : pyr.BH[3];
    // only the bricks the window really touches (2-3 columns, 3-4 rows) that lie inside the map: columns [clo, chi) and rows
    // [rlo, rhi) of the 3 x 4 block -> a 12-bit mask, bit 3 r + c
    const int nbx = (((wx0 & 7) + 9) >> 3) + 1, nby = (((wy0 & 3) + 9) >> 2) + 1;
    const int clo = max(0, -bx0), chi = max(min(nbx, BWl - bx0), clo);
    const int rlo = max(0, -by0), rhi = max(min(nby, BHl - by0), rlo);
    const unsigned cm = (1u << (chi & 31)) - (1u << (clo & 31));                                   // 0 when the range is empty
    const unsigned rm = 0x249u & ((1u << ((3 * rhi) & 31)) - (1u << ((3 * rlo) & 31)));            // bit 3 r for r in [rlo, rhi)
    P_m12 = (int)((clo < 3 && rlo < 4) ? rm * cm : 0u);                                            // (shift counts are in range then)
    // (unsigned: the product wraps for origins far outside the map — every brick of such a unit is masked)
    P_sb = (int)((unsigned)(by0 * BWl + bx0) * (unsigned)(LP * 128));
  }

  v4f bv[DEPTH][2];
  // lane -> brick bi = 8k + (lane >> 3) of the 3 x 4 block of bricks (bi < 12), 16-byte part lane & 7 of its line
  int bxi[2], byi[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) { const int bi = 8 * k + (lane >> 3); byi[k] = bi / 3; bxi[k] = bi - 3 * byi[k]; }
  // byte offset of this lane's part of brick (byi, bxi) from the window's first brick, per level
  int loff[4][2];
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int k = 0; k < 2; ++k) loff[l][k] = (byi[k] * pyr.BW[l] + bxi[k]) * (LP * 128) + (lane & 7) * 16;
  const int big = 0x7FFFFFF0;   // an offset past the end of every level: the load returns zeros
  auto rep8 = [](unsigned m8) __attribute__((always_inline)) {   // bit b of m8 -> bits 8 b .. 8 b + 7 (scalar unit)
    unsigned long long r;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"(m8));
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"((unsigned)r));
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"((unsigned)r));
    return r;
  };
  auto issue = [&](int pi, int sl) __attribute__((always_inline)) {   // sl = pi % DEPTH: ring slot of the unit; its level is sl & 3
    const int l = sl & 3;
    const int pp = min(pbase + (pi >> 2), np - 1);
    // (units past the wave's 16th — the prefetches behind the last one — read the copies in lanes 16 .. 18: valid, unused)
    const unsigned m12 = (unsigned)__builtin_amdgcn_readlane(P_m12, pi);
    const int sbase = (int)((unsigned)__builtin_amdgcn_readlane(P_sb, pi) + (unsigned)((poff + pp) * 128));
    const unsigned long long lm0 = rep8(m12 & 0xFFu), lm1 = rep8(m12 >> 8);
    // the pixel's map of this level as a buffer: a lane whose brick is outside the window or outside the map gets an offset
    // past the end, and the load returns zeros by itself (they ARE grid_sample's zero padding): no select on the data
    // (the pixel block's level as a buffer of NBK x 64 lines; the unit's pixel selects the line inside a brick's 8 KB run)
    const long NBl = pyr.NB[l];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pyr.base[l] + ((long)pair * pyr.NPB + pblk) * NBl * LP), 0, (int)(NBl * LP * 4), 0x00020000);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int off;
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(off) : "v"(big), "v"((int)((unsigned)loff[l][k] + (unsigned)sbase)), "s"(k == 0 ? lm0 : lm1));
      bv[sl][k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
  float* gw = grid[wave];
  // The unit body has no lane permutes (eight ds_bpermute per unit in round 2), no table round trip (rounds 3-4) and never
  // drains the LDS queue: it relies on what the hardware guarantees — the LDS instructions of ONE wave execute in issue order,
  // so a read issued after a write of the same wave sees it, and the next unit's grid writes cannot overtake this unit's reads.
  // per-lane constants: grid slot of this lane's two brick parts; its two samples (i, j), their cell inside the window and where
  // they go in a pixel's row of the A tile (the 47 lanes past the 81st sample compute sample (8, 8) again into the row's dump)
  int gofs[2], dofs[4][2], sofs[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int part = lane & 7;   // bricks 12..15 do not exist: their lanes load zeros and write them to rows 16..23
    gofs[k] = (byi[k] * 4 + (part >> 1)) * GW + bxi[k] * 8 + (part & 1) * 4;
    // sample (i, j) of this lane: pass 0 = the 8 x 8 block i, j < 8 (i = lane & 7: every 32-lane half reads 8 columns x 4
    // rows); pass 1 = column i = 8 (lanes 0-8) and row j = 8 (lanes 9-16)
    const int i9 = k == 0 ? (lane & 7) : (lane < 9 ? 8 : (lane < 17 ? lane - 9 : 8));
    const int j9 = k == 0 ? (lane >> 3) : (lane < 9 ? lane : 8);
    sofs[k] = (j9 * GW + i9) * 4;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int c = l * 81 + i9 * 9 + j9;
      dofs[l][k] = (k == 0 || lane < 17) ? (c >> 5) * 128 + (c & 31) * 2 : ADUMP + (lane & 15) * 2;
    }
  }
  auto fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // (compiler only: no instruction)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  static_assert(UPW % DEPTH == 0, "units per wave must be a multiple of the prefetch depth");
  float vmax = 0.f;   // largest sample magnitude of this lane: the saturation test of the sf format, once, after the loop
  // the unit loop is unrolled by DEPTH so that the register slots (= levels) are compile-time constants
  for (int pi0 = 0; pi0 < UPW; pi0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int pi = pi0 + d;
      issue(pi + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
      // window bricks -> grid. No wait: these writes come after the previous unit's sample reads and before this unit's
#pragma unroll
      for (int k = 0; k < 2; ++k) *reinterpret_cast<v4f*>(gw + gofs[k]) = bv[d][k];
      fence();
      // the unit's four bilinear products, scalars (nw, ne, sw, se of grid_sample)
      const float w00 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P_w00), pi));
      const float w01 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P_w01), pi));
      const float w10 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P_w10), pi));
      const float w11 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P_w11), pi));
      const char* q0 = reinterpret_cast<const char*>(gw) + __builtin_amdgcn_readlane(P_gb, pi);
      float q00[2], q01[2], q10[2], q11[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* q = reinterpret_cast<const float*>(q0 + sofs[t]);
        q00[t] = q[0]; q01[t] = q[1]; q10[t] = q[GW]; q11[t] = q[GW + 1];
      }
      char* arow = atile + (pbase + (pi >> 2)) * APITCH;
      float va[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // explicit FMAs in grid_sample's order (nw, ne, sw, se): a pixel's samples must not depend on its position in the
        // block or on the unrolled copy that computes them (the batch-invariance test)
#pragma clang fp contract(off)
        const float v = __builtin_fmaf(q11[t], w11, __builtin_fmaf(q10[t], w10, __builtin_fmaf(q01[t], w01, q00[t] * w00)));
        va[t] = v;
        // split: hi = f16(clamp(v)), lo = f16(v - hi) as one v_fma_mixlo_f16 (the same bits as convert-subtract-convert)
        const float vc = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
        unsigned hi, lo;
        asm("v_cvt_f16_f32_e32 %0, %1" : "=v"(hi) : "v"(vc));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(vc));
        // (no branch around the store: lanes past the 81st sample write to the row's dump bytes)
        unsigned short* dst = reinterpret_cast<unsigned short*>(arow + dofs[d & 3][t]);
        dst[0] = (unsigned short)hi;
        dst[32] = (unsigned short)lo;
      }
      vmax = fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), vmax);
    }
  }
  const bool clamped = !(vmax <= 65504.f);
  sf_report(clamped);

  // ---- phase 2: [TP pixels x 352] x convc1^T -> 256 channels on v_mfma_f32_16x16x32_f16. Wave w owns the CB 16-channel blocks
  // 16 (CB w + cb) .. + 15 for ALL TP pixels (PB 16-pixel column blocks): the block fetches the 352 x 256 weight matrix exactly
  // once (352 KiB per TP pixels), straight from L2 in operand order (weights.h: pack_fragment_major16, one contiguous KiB per wave
  // load) through a ring of WD chunks, the first WD requested BEFORE the barrier that ends the sampling phase; weights are the
  // row operand, so lane (n, g) ends up with channels 16 (CB w + cb) + 4 g + 0..3 of pixel 16 pb + n and stores 8 + 8 bytes.
  // (At TP = 32 the matrix is fetched once per 32 pixels — 5,600 cycles of the L1's 64 B/clk per block, the rate that bound the
  // 32-pixel block of round 2 — but now under the OTHER resident block's sampling phase, which needs the L1 for 100 KB of bricks.)
  constexpr int NQ = NCH / 32;
  constexpr int WD = 4;
  constexpr int CB = 256 / 16 / NWAVE, PB = TP / 16;
  static_assert(CB * NWAVE * 16 == 256, "the block's waves cover convc1's 256 output channels");
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int n16 = lane & 15, g16 = lane >> 4;
  const char* wbase = reinterpret_cast<const char*>(wfrag) + (long)(wave * CB) * NQ * 2048 + lane * 16;
  f16x8 wh[WD][CB], wl[WD][CB];
  auto load_w = [&](int slot, int q) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const char* p = wbase + ((long)cb * NQ + q) * 2048;
      wh[slot][cb] = *reinterpret_cast<const f16x8*>(p);
      if (!FAST) wl[slot][cb] = *reinterpret_cast<const f16x8*>(p + 1024);
    }
  };
  if (FUSED) {
#pragma unroll
    for (int q = 0; q < WD; ++q) load_w(q, q);
  }
  __syncthreads();

  if (!FUSED) {   // the sampled rows themselves: 352 channels = 1408 bytes per pixel, dwordx4 per thread
    for (int i = tid; i < np * 88; i += NWAVE * 64) {
      const int row = i / 88, part = i - row * 88;
      *reinterpret_cast<v4f*>(reinterpret_cast<char*>(out) + (p0 + row) * (NCH * 4L) + part * 16) =
          *reinterpret_cast<const v4f*>(atile + row * APITCH + part * 16);
    }
    return;
  }

  f32x4v acc[CB][PB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cb][pb][e] = 0.f;
  const char* arow = atile + n16 * APITCH + 16 * g16;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + 16 * pb * APITCH + q * 128);
      f16x8 al = ah;
      if (!FAST) al = *reinterpret_cast<const f16x8*>(arow + 16 * pb * APITCH + q * 128 + 64);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        f32x4v c = acc[cb][pb];
        if (!FAST) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q % WD][cb], al, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q % WD][cb], ah, c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q % WD][cb], ah, c, 0, 0, 0);
        acc[cb][pb] = c;
      }
    }
    if (q + WD < NQ) load_w(q % WD, q + WD);
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int c = 16 * (wave * CB + cb) + 4 * g16;
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int px = 16 * pb + n16;
      if (px < np) {
        const float4 o = make_float4(fmaxf(acc[cb][pb][0] * wscale + b.x, 0.f), fmaxf(acc[cb][pb][1] * wscale + b.y, 0.f),
                                     fmaxf(acc[cb][pb][2] * wscale + b.z, 0.f), fmaxf(acc[cb][pb][3] * wscale + b.w, 0.f));
        sf_store4(out + (p0 + px) * 256, 0, c, o);
      }
    }
  }
}

// feature rows in brick order: dst[img][n'][C] = src[img][y * W + x][C] for the cell (y, x) that brick position n'
// addresses, zero rows for the padding cells. One thread = 16 bytes (the sf chunk layout is copied verbatim).
__global__ void brick_rows_kernel(const float* __restrict__ src, long sb, int H, int W, int BW, int C,
                                  float* __restrict__ dst, long db, long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = C / 4;
  const int part = (int)(i % c4);
  const long row = i / c4;
  const long per_img = db / C;
  const long img = row / per_img;
  const int n = (int)(row - img * per_img);
  const int brick = n >> 5, within = n & 31;
  const int y = (brick / BW) * 4 + (within >> 3), x = (brick % BW) * 8 + (within & 7);
  v4f v = {0.f, 0.f, 0.f, 0.f};
  if (y < H && x < W) v = *reinterpret_cast<const v4f*>(src + img * sb + ((long)y * W + x) * C + part * 4);
  *reinterpret_cast<v4f*>(dst + img * db + (long)n * C + part * 4) = v;
}

// bricked level -> row-major [pixels][H * W] (debug reads)
__global__ void unbrick_kernel(const float* __restrict__ src, long NB, int N, int NPB, int H, int W, int BW, float* __restrict__ dst,
                               long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int hw = H * W;
  const long p = i / hw;
  const int c = (int)(i - p * hw);
  const int y = c / W, x = c - y * W;
  const long pair = p / N, pin = p - pair * N;
  const long brick = (long)(y >> 2) * BW + (x >> 3);
  dst[i] = src[(((pair * NPB + (pin >> 6)) * (NB / 32) + brick) * 64 + (pin & 63)) * 32 + (y & 3) * 8 + (x & 7)];
}

}  // namespace

void launch_brick_rows(const float* src, long sb, int nimg, int H, int W, int C, float* dst, long db, hipStream_t st) {
  ATDN_CHECK(C % 4 == 0 && db % C == 0, "brick_rows: bad geometry");
  const int BW = (W + 7) / 8;
  const long total4 = (long)nimg * (db / C) * (C / 4);
  hipLaunchKernelGGL(brick_rows_kernel, dim3((unsigned)cdivl(total4, 256)), dim3(256), 0, st, src, sb, H, W, BW, C, dst, db, total4);
  ATDN_HIP(hipGetLastError());
}

void launch_unbrick(const float* src, long NB, int N, int H, int W, long npix, float* dst, hipStream_t st) {
  const long total = npix * H * W;
  hipLaunchKernelGGL(unbrick_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, src, NB, N, brick_pixel_blocks(N), H, W,
                     (W + 7) / 8, dst, total);
  ATDN_HIP(hipGetLastError());
}

void launch_lookup_conv(const BrickPyramid& pyr, const float* coords1, long npix, float* coords_used, const float* wfrag,
                        float wscale, const float* bias, float* out, bool fast, hipStream_t st) {
  ATDN_CHECK(wfrag && bias && out, "lookup_conv: missing operand");
  ATDN_CHECK(pyr.N >= 1 && npix % pyr.N == 0, "lookup_conv: npix must be whole pairs of pyr.N pixels");
  const dim3 grid((unsigned)((npix / pyr.N) * cdiv(pyr.N, TP)));
  if (fast) hipLaunchKernelGGL((lookup_conv_kernel<true, true>), grid, dim3(NWAVE * 64), 0, st, pyr, coords1, npix, coords_used, wfrag, wscale, bias, out);
  else hipLaunchKernelGGL((lookup_conv_kernel<true, false>), grid, dim3(NWAVE * 64), 0, st, pyr, coords1, npix, coords_used, wfrag, wscale, bias, out);
  ATDN_HIP(hipGetLastError());
}

void launch_lookup_bricks(const BrickPyramid& pyr, const float* coords1, long npix, float* out, hipStream_t st) {
  ATDN_CHECK(pyr.N >= 1 && npix % pyr.N == 0, "lookup_bricks: npix must be whole pairs of pyr.N pixels");
  hipLaunchKernelGGL((lookup_conv_kernel<false, false>), dim3((unsigned)((npix / pyr.N) * cdiv(pyr.N, TP))), dim3(NWAVE * 64), 0, st, pyr, coords1,
                     npix, nullptr, nullptr, 1.f, nullptr, out);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
