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

constexpr int TP = 64;            // source pixels per block (round 3: 64 — the convc1 phase was bound by the L1 rate of its weight loads,
                                  // a block multiplies twice the pixels with every weight fragment it fetches)
constexpr int NCH = 352;          // 4 * 81 samples padded to a multiple of 32
constexpr int APITCH = 1440;      // bytes per pixel row of the A tile: 11 x 128 + 32 = 90 slots of 16 B, 90 = 2 (mod 4): conflict-free
                                  // ds_read_b128 for the operand lane map of v_mfma_f32_16x16x32_f16 (16 pixels x 4 slots per instruction)
constexpr int GW = 24, GH = 24;   // per-wave window grid: 3 x 4 bricks (24 x 16 cells). Pitch 24 dwords = -8 (mod 32 banks): the four
                                  // rows x two 16-byte halves a brick's eight lanes write (ds_write_b128: groups of 8 lanes) are eight
                                  // distinct bank quads, and a 32-lane group of sample reads (ds_read_b32) laid out as 8 x-positions x 4
                                  // rows is 32 distinct banks. (Pitch 28 with samples in channel order: SQ_LDS_BANK_CONFLICT = 25 % of
                                  // the kernel's LDS cycles.) Rows 16-23 are a dump for the lanes without a brick / a sample: the unit
                                  // body has NO branch (a branch around an LDS store made the compiler drain every prefetched load,
                                  // vmcnt(0), per unit)
constexpr int DEPTH = 4;          // (pixel, level) units in flight per wave
constexpr int NWAVE = 16;         // waves per block (one 1024-thread block per CU): wave w samples all four levels of pixels 4 w .. 4 w + 3
constexpr int UPW = TP * 4 / NWAVE;   // units (pixels of its level) per wave

// One (pixel, level) unit. Everything here is wave-uniform: the integers are forced into scalar registers
// (v_readfirstlane), so the brick range, the grid origin and the buffer descriptor of the unit are computed on the scalar
// unit and cost the vector pipe nothing (SQ counters of the round-3 kernel: the sampling phase is VALU-issue bound,
// ~180 vector instructions per unit; this and the chain table below took it to ~95).
struct Unit { float xc, yc; int sane; int wx0, wy0, bx0, by0; };

__device__ __forceinline__ Unit unit_origin(float cx, float cy, float inv) {
  Unit u;
  u.xc = cx * inv; u.yc = cy * inv;
  const bool sane = (fabsf(u.xc) < 1.0e6f) && (fabsf(u.yc) < 1.0e6f);   // also rejects NaN
  u.sane = __builtin_amdgcn_readfirstlane((int)sane);
  const int fx = __builtin_amdgcn_readfirstlane((int)floorf(u.xc)), fy = __builtin_amdgcn_readfirstlane((int)floorf(u.yc));
  u.wx0 = u.sane ? fx - 5 : -(1 << 24);
  u.wy0 = u.sane ? fy - 5 : -(1 << 24);
  u.bx0 = u.wx0 >> 3; u.by0 = u.wy0 >> 2;                       // arithmetic shifts: floor for negative origins
  return u;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// FUSED: multiply the sampled tile with convc1 and store relu(. + bias) as sf rows [pixel][256];
// !FUSED: store the sampled tile itself as sf rows [pixel][352] (debug reads, the unfused comparison path)
template <bool FUSED, bool FAST>
__global__ __launch_bounds__(NWAVE * 64, 1) void lookup_conv_kernel(const BrickPyramid pyr, const float* __restrict__ coords1,
                                                            const long npix, float* __restrict__ coords_used,
                                                            const float* __restrict__ wfrag, const float wscale,
                                                            const float* __restrict__ bias, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char atile[TP * APITCH];
  __shared__ __attribute__((aligned(16))) float grid[NWAVE][GH * GW];

  __shared__ __attribute__((aligned(16))) float2 ctab[NWAVE][2][64];   // per wave, two units: (weight, grid index) of the 18 chains
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (tells the compiler what the hardware guarantees: wave-uniform)
  // block = one pixel block of one pair (the last block of a pair is partial when N % 64 != 0)
  static_assert(TP == kBrickPixelBlock, "a block of this kernel is one pixel block of the bricked pyramid");
  const int bpp = (pyr.N + TP - 1) / TP;                        // blocks per pair
  const int pair = (int)blockIdx.x / bpp, pblk = (int)blockIdx.x - pair * bpp;
  const long p0 = (long)pair * pyr.N + (long)pblk * TP;         // first pixel of the block in coords1 / out
  const int np = min(TP, pyr.N - pblk * TP);
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
  static_assert(DEPTH == 4, "unit pi has level pi & 3 = pi % DEPTH");
  const int pbase = wave * (UPW / 4);
  float wm1[4], hm1[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { wm1[l] = (float)(pyr.W[l] - 1); hm1[l] = (float)(pyr.H[l] - 1); }
  // coordinates of the block's pixels: lane i holds pixel i (pixels past the end repeat the last one; never stored)
  static_assert(TP == 64, "one lane per pixel of the block");
  const long pc = p0 + min(lane, np - 1);
  const float2 cmine = *reinterpret_cast<const float2*>(coords1 + pc * 2);
  if (wave == 0 && lane < np && coords_used) *reinterpret_cast<float2*>(coords_used + (p0 + lane) * 2) = cmine;

  Unit un[DEPTH];
  v4f bv[DEPTH][2];
  // lane -> brick bi = 8k + (lane >> 3) of the 3 x 4 block of bricks (bi < 12), 16-byte part lane & 7 of its line
  int bxi[2], byi[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) { const int bi = 8 * k + (lane >> 3); byi[k] = bi / 3; bxi[k] = bi - 3 * byi[k]; }
  auto issue = [&](int pi, int l) __attribute__((always_inline)) {   // l = pi & 3: level of the unit AND its ring slot
    const int pp = min(pbase + (pi >> 2), np - 1);
    // (pp is wave-uniform: v_readlane, not a ds_bpermute whose wait would also sit behind the sampling phase's LDS traffic)
    const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cmine.x), pp));
    const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cmine.y), pp));
    un[l] = unit_origin(cx, cy, 1.0f / (float)(1 << l));
    const Unit& u = un[l];
    const int BWl = pyr.BW[l], BHl = pyr.BH[l];
    const long NBl = pyr.NB[l];
    // only the bricks the 12 x 12 window really touches: 2-3 columns, 3-4 rows of bricks (scalar arithmetic)
    const int nbx = (((u.wx0 & 7) + 11) >> 3) + 1, nby = (((u.wy0 & 3) + 11) >> 2) + 1;
    // the pixel's map of this level as a buffer: a lane whose brick is outside the window or outside the map gets an offset
    // past the end, and the load returns zeros by itself (they ARE grid_sample's zero padding): no select on the data
    // (the pixel block's level as a buffer of NBK x 64 lines; the unit's pixel selects the line inside a brick's 8 KB run)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pyr.base[l] + ((long)pair * pyr.NPB + pblk) * NBl * TP), 0, (int)(NBl * TP * 4), 0x00020000);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int bx = u.bx0 + bxi[k], by = u.by0 + byi[k];
      const bool ok = (bxi[k] < nbx) & (byi[k] < nby) & ((unsigned)bx < (unsigned)BWl) & ((unsigned)by < (unsigned)BHl);
      const int off = ok ? ((by * BWl + bx) * TP + pp) * 128 + (lane & 7) * 16 : 0x7FFFFFF0;
      bv[l][k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
  float* gw = grid[wave];
  bool clamped = false;   // saturation of the sf format, reported once after the loop (sf.h)
  // The unit body has no lane permutes (eight ds_bpermute per unit in round 2) and never drains the LDS queue (twice per unit
  // in round 2): it relies on what the hardware guarantees — the LDS instructions of ONE wave execute in issue order, so a
  // read issued after a write of the same wave sees it, and the next unit's grid writes cannot overtake this unit's reads.
  // The 9 + 9 coordinate chains of a unit (x offsets -4..4, y offsets -4..4; the reference's arithmetic, corr.py:43-49 and
  // utils.py:63-70) are evaluated ONCE, by lanes 0-17, one unit ahead, and left in a 64-entry table of the wave; a sample
  // lane picks up the (weight, grid index) pairs of its two samples with four ds_read_b64 — LDS instructions, while the
  // vector pipe does the arithmetic of the current unit.
  // per-lane constants: grid slot of this lane's two brick parts; its two samples (i, j), their table entries and where they
  // go in a pixel's row of the A tile (the 47 lanes past the 81st sample compute sample (8, 8) again into the dump rows)
  int gofs[2], dofs[4][2], ex[2], ey[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int part = lane & 7;   // bricks 12..15 do not exist: their lanes load zeros and write them to rows 16..23
    gofs[k] = (byi[k] * 4 + (part >> 1)) * GW + bxi[k] * 8 + (part & 1) * 4;
    // sample (i, j) of this lane: pass 0 = the 8 x 8 block i, j < 8 (i = lane & 7: every 32-lane half reads 8 columns x 4
    // rows); pass 1 = column i = 8 (lanes 0-8) and row j = 8 (lanes 9-16)
    const int i9 = k == 0 ? (lane & 7) : (lane < 9 ? 8 : (lane < 17 ? lane - 9 : 8));
    const int j9 = k == 0 ? (lane >> 3) : (lane < 9 ? lane : 8);
    ex[k] = i9; ey[k] = 9 + j9;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int c = l * 81 + i9 * 9 + j9;
      dofs[l][k] = (c >> 5) * 128 + (c & 31) * 2;
    }
  }
  char* const dump = reinterpret_cast<char*>(gw + 22 * GW) + (lane & 15) * 2;
  // chain constants of THIS lane's table entry: lanes 0-8 x offset lane - 4, lanes 9-17 y offset lane - 13 (the rest: unused entries)
  const bool isx = lane < 9;
  const float cfd = (float)((isx ? lane : min(lane - 9, 8)) - 4);
  float csz[4], crs[4], chs[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { csz[l] = isx ? wm1[l] : hm1[l]; crs[l] = 1.0f / csz[l]; chs[l] = csz[l] / 2.f; }   // (IEEE divisions, once per wave)
  float2* const tab = ctab[wave][0];
  // pos = c + d; g = 2 pos / (S - 1) - 1; u = (g + 1) * ((S - 1) / 2); weight = u - floor(u);
  // grid index = clamp(floor(u) - window origin, 0, 10) + origin in the grid. The division is a multiplication by the correctly
  // rounded reciprocal plus one FMA correction step. That is the correctly rounded quotient whenever the first product is
  // within one ulp of it (Markstein) — not a theorem for every (t, b); for the divisors this network has (W_l - 1, H_l - 1 of
  // the four levels at C1 and C2) tests/test_host_arith.py checks the chain against the IEEE division over every position on
  // a 1/64-pixel lattice across the maps plus 16 pixels of margin, so "the same bits as the division" holds where it is used.
  auto chain_to_table = [&](const Unit& u, int l, int buf) __attribute__((always_inline)) {
#pragma clang fp contract(off)   // the same roundings in every unrolled copy (results must not depend on a pixel's slot)
    const float c0 = isx ? u.xc : u.yc;
    const int org = isx ? u.wx0 : u.wy0, gorg = isx ? u.wx0 - 8 * u.bx0 : u.wy0 - 4 * u.by0;
    const float t = 2.f * (c0 + cfd);
    float q = t * crs[l];
    q = __builtin_fmaf(__builtin_fmaf(-q, csz[l], t), crs[l], q);          // = t / csz (see above)
    const float uu = ((q - 1.f) + 1.f) * chs[l];
    const float fl = floorf(uu);
    // (a unit that is not sane has its origin at -2^24: the clamp alone keeps the index inside the grid)
    const int idx = min(max((int)fl - org, 0), 10) + gorg;
    tab[buf * 64 + lane] = make_float2(uu - fl, __int_as_float(idx));
  };
  struct Taps { float ww[2], nn[2]; int off[2]; };   // per pass: x weight, y weight, grid offset (floats) of the top-left tap
  auto read_table = [&](int buf, Taps& tp) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float2 x = tab[buf * 64 + ex[t]], y = tab[buf * 64 + ey[t]];
      tp.ww[t] = x.x; tp.nn[t] = y.x;
      tp.off[t] = __float_as_int(y.y) * GW + __float_as_int(x.y);
    }
  };
  auto fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // (compiler only: no instruction)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  static_assert(UPW % DEPTH == 0, "units per wave must be a multiple of the prefetch depth");
  Taps tcur, tnext;
  chain_to_table(un[0], 0, 0);
  fence();
  read_table(0, tcur);
  // the unit loop is unrolled by DEPTH so that the register slots (= levels) are compile-time constants
  for (int pi0 = 0; pi0 < UPW; pi0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int pi = pi0 + d;
      issue(pi + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
      const Unit u = un[d];
      // window bricks -> grid. No wait: these writes come after the previous unit's sample reads and before this unit's
#pragma unroll
      for (int k = 0; k < 2; ++k) *reinterpret_cast<v4f*>(gw + gofs[k]) = bv[d][k];
      fence();
      float q00[2], q01[2], q10[2], q11[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* q = gw + tcur.off[t];
        q00[t] = q[0]; q01[t] = q[1]; q10[t] = q[GW]; q11[t] = q[GW + 1];
      }
      // the next unit's chains -> the other half of the table, and this lane's entries of it back (its coordinates arrived
      // DEPTH - 2 units ago); both behind the sample reads in the LDS queue
      chain_to_table(un[(d + 1) % DEPTH], (d + 1) % DEPTH, (d + 1) & 1);
      fence();
      read_table((d + 1) & 1, tnext);
      char* arow = atile + (pbase + (pi >> 2)) * APITCH;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // explicit FMAs: left to itself the compiler fused these products differently in different copies of the unrolled
        // loop, and a pixel's samples depended on its position in the block (the batch-invariance test caught it)
#pragma clang fp contract(off)
        const float ww = tcur.ww[t], nn = tcur.nn[t];
        const float ee = 1.f - ww, ss = 1.f - nn;
        float v = __builtin_fmaf(q11[t], ww * nn, __builtin_fmaf(q10[t], ee * nn, __builtin_fmaf(q01[t], ww * ss, q00[t] * (ee * ss))));
        v = u.sane ? v : 0.f;
        const SfPair sp = sf_split_flag(v, clamped);
        // (no branch around the store: lanes past the 81st sample write to the dump rows)
        _Float16* dst = reinterpret_cast<_Float16*>((t == 0 || lane < 17) ? arow + dofs[d][t] : dump);
        dst[0] = sp.hi;
        dst[32] = sp.lo;
      }
      tcur = tnext;
    }
  }
  sf_report(clamped);

  // ---- phase 2: [64 pixels x 352] x convc1^T -> 256 channels on v_mfma_f32_16x16x32_f16. Wave w owns the 16 channels
  // 16 w .. 16 w + 15 for ALL 64 pixels (four 16-pixel column blocks), so the block fetches the 352 x 256 weight matrix
  // exactly once (352 KiB per 64 pixels; the 8-wave / 32-pixel block of round 2 fetched it once per 32 pixels and was
  // bound by the L1 rate of those loads: 5,600 cycles at 64 B/clk against 4,200 cycles of MFMA per block). Weights come
  // straight from L2 in operand order (weights.h: pack_fragment_major16, one contiguous KiB per wave load) through a ring
  // of WD chunks, the first WD requested BEFORE the barrier that ends the sampling phase; weights are the row operand, so
  // lane (n, g) ends up with channels 16 w + 4 g + 0..3 of pixel 16 pb + n and stores 8 + 8 bytes.
  constexpr int NQ = NCH / 32;
  constexpr int WD = 4;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int n16 = lane & 15, g16 = lane >> 4;
  const char* wbase = reinterpret_cast<const char*>(wfrag) + (long)wave * NQ * 2048 + lane * 16;
  f16x8 wh[WD], wl[WD];
  auto load_w = [&](int slot, int q) __attribute__((always_inline)) {
    const char* p = wbase + (long)q * 2048;
    wh[slot] = *reinterpret_cast<const f16x8*>(p);
    if (!FAST) wl[slot] = *reinterpret_cast<const f16x8*>(p + 1024);
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

  f32x4v acc[4];   // [pixel block]
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[pb][e] = 0.f;
  const char* arow = atile + n16 * APITCH + 16 * g16;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + 16 * pb * APITCH + q * 128);
      f32x4v c = acc[pb];
      if (!FAST) {
        const f16x8 al = *reinterpret_cast<const f16x8*>(arow + 16 * pb * APITCH + q * 128 + 64);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q % WD], al, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[q % WD], ah, c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[q % WD], ah, c, 0, 0, 0);
      acc[pb] = c;
    }
    if (q + WD < NQ) load_w(q % WD, q + WD);
  }
  {
    const int c = 16 * wave + 4 * g16;
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const int px = 16 * pb + n16;
      if (px < np) {
        const float4 o = make_float4(fmaxf(acc[pb][0] * wscale + b.x, 0.f), fmaxf(acc[pb][1] * wscale + b.y, 0.f),
                                     fmaxf(acc[pb][2] * wscale + b.z, 0.f), fmaxf(acc[pb][3] * wscale + b.w, 0.f));
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
