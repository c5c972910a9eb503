// All-pairs correlation, one pyramid level (corr.py:55-63: corr[p, q] = <fmap1[:, p], fmap2[:, q]> / sqrt(256)), written
// straight into the bricked layout the lookup reads (kernels.h: BrickPyramid: pixel blocks of 64 source pixels, brick-major
// inside), i.e. a plain NT-GEMM  out = scale * F1 (N x 256) * F2b^T (NB x 256)  whose B operand is the target feature map in
// brick order (brick_rows_kernel; zero rows for padding cells; column n' = brick n' / 32, cell n' % 32) and whose output tiles
// of 32 pixels x one brick are 4 KB contiguous. Levels 1-3 use 2 x 2-pooled target features
// (the average of corr.py:28-30 commutes with the dot product).
//
// Round 4: this replaces the generic GEMM kernel (conv_sf.h: 128 x 128 tiles, BOTH operands staged through LDS per 32-deep
// chunk, two block barriers per chunk) for this shape. Level 0 at 16 pairs: 455 GFLOP x 3 and 3.56 GB of output; the generic
// kernel took 1.56-1.64 ms = 0.11 of the f16 MFMA peak and fetched 3.8 GB for < 0.25 GB of unique operands. Here
//   * a block owns 128 source pixels, one 32-pixel strip per wave, and the strip's WHOLE K = 256 (hi | lo, 32 KB) stays in
//     registers for the sweep, as qk_softmax_kernel keeps its queries (attention.hip);
//   * the target rows are streamed once per block in tiles of 32 rows x 256 channels (32 KB) through a double-buffered LDS
//     image (one barrier per tile; the tile after next is in flight in registers meanwhile);
//   * targets are the MFMA row operand, so a lane ends up with FOUR consecutive brick cells of ONE source pixel; a tile of 32
//     target rows is ONE brick = one 128-byte line per source pixel, and adjacent lanes swap one accumulator block by DPP so that
//     every 16-byte non-temporal store instruction writes WHOLE lines (half-line stores run at half the rate: see the kernel);
//   * the brick-column range is cut into `splits` slices (blocks of one pixel strip x one slice) so that the grid fills
//     whole rounds of the 2 x 256 resident blocks.
#include "kernels.h"

#include "conv_mfma.h"
#include "sf.h"

namespace atdn {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4f gv4f;   // explicitly global: a loop-carried pointer came out as flat_store

struct CorrArgs {
  const float* f1; long sb1;     // sf [pair][N][256]
  const float* f2b; long sb2;    // sf [pair][NB][256], brick order
  float* out;                    // fp32 [pair][pixel block][NB / 32][64][32] (kernels.h: BrickPyramid)
  int B, N, NB, NPB, strips, splits, tiles_per_split;
  float scale;
};

template <bool FAST>
__global__ __launch_bounds__(256, 2) void corr_bricks_kernel(const CorrArgs a) {
  constexpr int KROW = 160;            // LDS pitch of a target row: conflict-free ds_read_b128 for the 16x16x32 lane map
  constexpr int KT = 32;               // target rows per LDS tile
  constexpr int NC = 8;                // 32-channel chunks of K = 256
  constexpr int CH = KT * KROW;        // one chunk of a tile
  constexpr int IMG = NC * CH;         // [8 chunks][32 rows][160 B] = 40 KB
  __shared__ __attribute__((aligned(16))) char lds[2 * IMG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = a.B * a.splits * a.strips;
  const int id = xcd_remap(blockIdx.x, nblk);          // strips of one (pair, slice) are neighbours: they share the slice's rows in L2
  const int strip4 = id % a.strips;
  const int rest = id / a.strips;
  const int split = rest % a.splits, b = rest / a.splits;
  const int n16 = lane & 15, g16 = lane >> 4;
  const char* f1 = reinterpret_cast<const char*>(a.f1 + (long)b * a.sb1);
  const char* f2 = reinterpret_cast<const char*>(a.f2b + (long)b * a.sb2);

  // source-pixel fragments (the column operand), resident for the whole sweep: [16-pixel block][channel chunk]
  f16x8 qh[2][NC], ql[2][NC];
  int mrow[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    mrow[rb] = (strip4 * 4 + wave) * 32 + 16 * rb + n16;
    const char* qrow = f1 + (long)min(mrow[rb], a.N - 1) * 1024 + 16 * g16;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      qh[rb][c] = *reinterpret_cast<const f16x8*>(qrow + c * 128);
      if (!FAST) ql[rb][c] = *reinterpret_cast<const f16x8*>(qrow + c * 128 + 64);
    }
  }

  // tile loader: thread -> target row lr of the tile, 16-byte slot ls of every 128-byte channel chunk
  const int lr = tid >> 3, ls = tid & 7;
  const int t0 = split * a.tiles_per_split;
  const int nt = min(a.tiles_per_split, a.NB / KT - t0);   // (>= 1: the host never launches an empty slice)
  v4f kreg[NC];
  // (Round 4 also built these loads as inline asm with an exactly counted wait, vmcnt(4), before a tile's LDS writes — with loads
  // AND stores pending the compiler's wait-count pass drains everything, vmcnt(0), i.e. it also waits for the previous
  // iteration's stores. Measured in one job: no difference, 1.19-1.23 ms either way; the compiler-counted form stays.)
  auto fetch = [&](int j) __attribute__((always_inline)) {
    const char* src = f2 + (long)((t0 + j) * KT + lr) * 1024 + 16 * ls;
#pragma unroll
    for (int c = 0; c < NC; ++c) kreg[c] = *reinterpret_cast<const v4f*>(src + c * 128);
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NC; ++c) *reinterpret_cast<v4f*>(lds + buf * IMG + c * CH + lr * KROW + 16 * ls) = kreg[c];
  };
  // Result stores. After the MFMAs lane (n, g) holds cells 16 kb + 4 g + 0..3 of pixel 16 rb + n: stored as they stand, one
  // instruction writes 16 HALF lines (64 B per pixel) and the wave's next instruction the other halves — and half-line stores run
  // at half the rate of whole-line stores on this chip (tools/microbench_stream.py: 2.6 against 4.6-5.2 TB/s; the first version
  // of this kernel was bound by exactly that: 1.31 ms with its stores, 0.97 ms without). So adjacent lanes (pixels n, n ^ 1)
  // swap one accumulator block by DPP: instruction 1 writes the even pixels' WHOLE lines (the even lane its own cells 0-15, the
  // odd lane its partner's cells 16-31), instruction 2 the odd pixels'.
  // Stores are UNCONDITIONAL (vmcnt retires in order and counts stores: the wait for the tile loads is "all but the four
  // youngest" only if those four always exist): a pair's region holds whole 128-pixel strips (NPB is even), so the lanes of
  // source rows past N write scratch rows of their own pair that nobody reads. The store is explicitly global and addressed by
  // an offset from ONE base pointer: a loop-carried pointer came out as flat_store, which counts on lgkmcnt too — the
  // barrier's LDS wait would then wait for HBM writes.
  const bool odd = (lane & 1) != 0;
  long orow[2][2];    // [16-pixel block][store]: offset from a.out, in floats
  const long NBK = a.NB / KT;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int prow = h == 0 ? (mrow[rb] & ~1) : (mrow[rb] | 1);   // the pixel whose line this lane helps to write
      orow[rb][h] = ((((long)b * a.NPB + (prow >> 6)) * NBK + t0) * 64 + (prow & 63)) * KT + (odd ? 16 : 0) + 4 * g16;
    }

  fetch(0);
  stash(0);
  __syncthreads();
  fetch(min(1, nt - 1));
  for (int j = 0; j < nt; ++j) {
    const char* img = lds + (j & 1) * IMG + n16 * KROW + 16 * g16;
    f32x4v acc[2][2];   // [target block of 16][pixel block of 16]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[kb][rb][e] = 0.f;
    // target fragments of chunk c + 1 are requested before the MFMAs of chunk c are issued (two register sets)
    f16x8 kh[2][2], kl[2][2];   // [set][target block]
    auto read_rows = [&](int set, int c) __attribute__((always_inline)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const char* kp = img + c * CH + 16 * kb * KROW;
        kh[set][kb] = *reinterpret_cast<const f16x8*>(kp);
        if (!FAST) kl[set][kb] = *reinterpret_cast<const f16x8*>(kp + 64);
      }
    };
    read_rows(0, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) read_rows((c + 1) & 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          f32x4v v = acc[kb][rb];
          if (!FAST) {
            v = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[c & 1][kb], qh[rb][c], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[c & 1][kb], ql[rb][c], v, 0, 0, 0);
          }
          v = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[c & 1][kb], qh[rb][c], v, 0, 0, 0);
          acc[kb][rb] = v;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next tile goes into the other image (last read in iteration j - 1; every wave has passed the barrier since); the
    // tile after that is requested now and lands during the stores and the next MFMA block
    stash((j + 1) & 1);
    fetch(min(j + 2, nt - 1));
    // lane (n, g): source pixel 16 rb + n, brick cells 16 kb + 4 g + 0..3 of this tile (see above)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const f32x4v A = acc[0][rb] * a.scale, Bv = acc[1][rb] * a.scale;
      f32x4v recv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float send = odd ? A[e] : Bv[e];
        recv[e] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
      }
      f32x4v v1, v2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v1[e] = odd ? recv[e] : A[e]; v2[e] = odd ? Bv[e] : recv[e]; }
      __builtin_nontemporal_store(v1, reinterpret_cast<gv4f*>(reinterpret_cast<uintptr_t>(a.out + orow[rb][0])));
      __builtin_nontemporal_store(v2, reinterpret_cast<gv4f*>(reinterpret_cast<uintptr_t>(a.out + orow[rb][1])));
      orow[rb][0] += 64 * KT;   // the next brick of this pixel block: 64 pixels x 32 cells further
      orow[rb][1] += 64 * KT;
    }
    __syncthreads();
  }
}

}  // namespace

void launch_corr_bricks(const float* f1, long sb1, const float* f2b, long sb2, int B, int N, int NB, float scale, float* out,
                        bool fast, hipStream_t st) {
  ATDN_CHECK(f1 && f2b && out && B >= 1 && N >= 1 && NB >= 32 && NB % 32 == 0, "corr_bricks: bad argument");
  CorrArgs a;
  a.NPB = brick_pixel_blocks(N);
  a.f1 = f1; a.sb1 = sb1; a.f2b = f2b; a.sb2 = sb2; a.out = out; a.B = B; a.N = N; a.NB = NB; a.scale = scale;
  a.strips = cdiv(N, 128);
  const int tiles = NB / 32;
  // slices of the brick-column range: the count (<= 8, at least 8 tiles per slice) whose grid wastes least of its last round
  // of 512 resident blocks (2 per CU); ties go to fewer slices (the source strip is loaded once per block)
  int best = 1;
  double best_eff = 0.0;
  for (int s = 1; s <= 8; ++s) {
    const int tps = cdiv(tiles, s);
    if (s > 1 && tps < 8) break;
    const int used = cdiv(tiles, tps);             // slices that are not empty
    const double blocks = (double)B * a.strips * used;
    const double eff = blocks / (512.0 * std::ceil(blocks / 512.0));
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  a.tiles_per_split = cdiv(tiles, best);
  a.splits = cdiv(tiles, a.tiles_per_split);
  const int nblk = B * a.splits * a.strips;
  if (fast) hipLaunchKernelGGL(corr_bricks_kernel<true>, dim3(nblk), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(corr_bricks_kernel<false>, dim3(nblk), dim3(256), 0, st, a);
  ATDN_HIP(hipGetLastError());
}

}  // namespace atdn
