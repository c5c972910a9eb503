// GMA attention on the split-f16 engine (attention.hip): the [N x N] attention matrix of gma.py:54-76 is produced by a
// QK^T kernel with the row softmax fused in, stored in MFMA-operand ("fragment-major") order, and streamed by the
// attention x V kernel of gma.py:102-115 straight into operand registers.
#pragma once
#include "common.h"

namespace atdn {

// Geometry of the stored attention matrix of one batch.
//   strips  RT = ceil(N / 32) 32-row strips per pair, chunks Q = ldN / 32 32-column chunks per row
//   block (pair, strip, chunk) = 32 rows x 32 columns, strip-major: ((pair*RT + strip)*Q + chunk) * blk bytes
//   a block holds two ROW BLOCKS rb = 0, 1 of 16 rows each; lane (n = lane & 15, g = lane >> 4) owns, in row block rb, row
//   32*strip + 16*rb + n and the eight columns 32*chunk + 4*g + (i & 3) + 16*(i >> 2), i = 0..7 — the B operand of
//   v_mfma_f32_16x16x32_f16 (round 3; K = 32 = the whole chunk), with the columns of a key group in the accumulator
//   order of the producing MFMAs (two 16-key blocks), so producer stores and consumer loads need no shuffle; the V^T
//   operand is brought into the same column order when it is written to LDS.
// Values are e = exp(s - rowmax~) * 2^AT_SHIFT, NOT normalised: rinv[pair][row] = 1 / sum_k e is applied by the
// consumer. rowmax~ comes from a cheap first pass (f16 x f16 logits); softmax is shift-invariant, so any shift close to
// the true maximum gives the same probabilities and keeps e inside the f16 range (largest element of a row ~ 2^10).
// Element format H3, 3 bytes: hi = f16(e), residual as ONE BYTE in units of the group's ulp / 256 (group = a lane's eight
//   values of one k-step): e = hi + (byte - 128) * 2^(E - 33), E = f16 exponent of the group's largest hi.
//   block = [rb][lane][16 B] hi, then [lane][rb][8 B] residual bytes (both row blocks of a lane adjacent: one 16-byte access) = 3072 B.
//   H3 carries 19 significant bits of every value within 2^-8 of its group's maximum and an ABSOLUTE error below
//   2^-20 of the group maximum everywhere — what matters for sum_k e_k v_k with fp32 accumulation (the residual of a
//   4-byte hi | lo pair is an f16 subnormal for everything below 2^-4 of the row maximum, i.e. no better) — and moves 25 %
//   fewer bytes through the kernel that streams the matrix twelve times. (The 4-byte format was removed in round 3.)
//   A value past the f16 range is clamped to 65504 and counted by the sf saturation counter (sf.h).
constexpr int AT_SHIFT = 10;
constexpr int AT_BLK_BYTES = 3072;
struct AttnGeom {
  int B, N, ldN, RT, Q, Npad;   // Npad = 32 * RT: row count of rowmax / rinv per pair
};
inline AttnGeom attn_geom(int B, int N, int ldN) {
  return {B, N, ldN, (N + 31) / 32, ldN / 32, ((N + 31) / 32) * 32};
}
inline long attn_floats(const AttnGeom& g) { return (long)g.B * g.RT * g.Q * (AT_BLK_BYTES / 4); }

// qk: sf [B][N][256] = q (pre-scaled, channels 0..127) | k (channels 128..255)            (gma.py:57-60)
// pass 1: rowmax[b][m] ~ max_n q_m . k_n  (f16 x f16 products)
void launch_qk_rowmax(const float* qk, const AttnGeom& g, float* rowmax, hipStream_t st);
// pass 2: P (fragment-major, see above) and rinv[b][m] = 1 / sum_n exp(s_mn - rowmax_m) * 2^-AT_SHIFT scaling included
void launch_qk_softmax(const float* qk, const AttnGeom& g, const float* rowmax, float* P, float* rinv, bool fast,
                       hipStream_t st);
// out[b][m][c] = mf[b][m][c] + gamma * rinv[b][m] * sum_k P[b][m][k] * V[b][k][c]          (gma.py:111-115)
//   vT sf [B][128][ldN] (V transposed, k contiguous, the 32 keys of every chunk in the operand order above: SfVT);
//   mf / out sf rows with pixel stride ld (floats), per-pair stride sb
// `part` != nullptr (the low-latency form, round 6): when the launch would leave most CUs idle (B <= 4 at KITTI size) the key axis
// is cut into attn_v_splits(g) ranges, one block each, with fp32 partial slabs in `part` ([8][B][Npad][128] floats) and a reduce
// pass; another summation order than the one-block-per-tile form, so only callers that asked for it get it.
int attn_v_splits(const AttnGeom& g);
void launch_attn_v(const float* P, const float* rinv, const AttnGeom& g, const float* vT, const float* gamma,
                   const float* mf, float* out, long sb, int ld, bool fast, hipStream_t st, float* part = nullptr);
// debug / tests: normalised probabilities as fp32 rows [B][N][ldN]
void launch_attn_decode(const float* P, const float* rinv, const AttnGeom& g, float* rows, hipStream_t st);

}  // namespace atdn
