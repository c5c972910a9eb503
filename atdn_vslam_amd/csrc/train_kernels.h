// Kernels of the CLVO training step that are not convolutions (clvo_train.hip): train-mode BatchNorm + Mish forward
// and backward on NHWC16 maps with per-call statistics groups, zero-stuffing for transposed convolutions, weight
// gradients of the thin convolutions, small dense GEMMs, LSTM cell forward/backward, loss and AdamW.
// Reference semantics: torch.nn.BatchNorm2d (training), nn.Mish, nn.LSTMCell, odometry/loss.py, torch.optim.AdamW.
#pragma once
#include <hip/hip_runtime.h>

namespace atdn {

// ---- weight packing for the ROW-mode conv engine, on the device (weights change every iteration)
// forward:   dst[n][ky*ldr + kx*Cpix + c] = w[n][c][ky][kx]                         rows = N
// transposed (for the data gradient): dst[c][ky*ldr + kx*Cpix + n] = w[n][c][KH-1-ky][KW-1-kx]   rows = Cin
// ldr = round_up(KW*Cpix, 32); every other entry of dst is written as 0.
void launch_pack_row(const float* w, int N, int Cin, int Cpix, int KH, int KW, bool transposed, float* dst, hipStream_t st);

// ---- train-mode BatchNorm over NHWC16 maps; G statistic groups of P pixels each (a group = one forward() call)
constexpr int BN_C = 16;
int bn_partial_blocks(long P);  // blocks per group the reduction kernels use
// part[G][nblk][2][16]: sum and sum of squares of a = mish(z) (mish=true) or of z itself
void launch_bn_stats(const float* z, int G, long P, bool mish, float* part, hipStream_t st);
// mean/rstd [G][16]; running stats updated sequentially over the G calls (momentum 0.1, unbiased variance)
// (var_scratch: [G][16] floats, the groups' unbiased variances between the two launches)
void launch_bn_finalize(const float* part, int G, long P, float* running_mean, float* running_var, float* mean,
                        float* rstd, float* var_scratch, hipStream_t st);
// the same with the partial rows counted by the caller (statistics taken in a producing kernel: Conv16Stats::rows, or
// bn_apply_partial_rows(P) after launch_bn_apply(..., next_part))
void launch_bn_finalize_rows(const float* part, int G, int rows, long P, float* running_mean, float* running_var, float* mean,
                             float* rstd, float* var_scratch, hipStream_t st);
// y = (act(z) - mean) * rstd * gamma + beta (+ add), act = mish or identity.
// next_part (optional, [G][bn_apply_partial_rows(P)][2][16]): sums of Mish(y), Mish(y)^2 — the statistics of a BatchNorm that
// follows y directly (ResidualConv.out_block: bn(mish(x + skip)), layers/conv.py:78-80,88)
int bn_apply_partial_rows(long P);
void launch_bn_apply(const float* z, int G, long P, bool mish, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const float* add, float* y, hipStream_t st, float* next_part = nullptr);
// backward: part[G][nblk][2][16] = sum dy, sum dy*xhat
void launch_bn_bwd_stats(const float* dy, const float* z, int G, long P, bool mish, const float* mean, const float* rstd,
                         float* part, hipStream_t st);
// sums[G][2][16] from the partials; dgamma += sum_g sum dy*xhat, dbeta += sum_g sum dy
void launch_bn_bwd_finalize(const float* part, int G, long P, float* sums, float* dgamma, float* dbeta, hipStream_t st);
// dz = gamma*rstd*(dy - sum_dy/P - xhat*sum_dyx/P) * act'(z); part_db[G][nblk][16] = per-block sums of dz
void launch_bn_bwd_apply(const float* dy, const float* z, int G, long P, bool mish, const float* mean, const float* rstd,
                         const float* gamma, const float* sums, float* dz, float* part_db, hipStream_t st);
// out[16] += sum over G*nblk partial rows of 16
void launch_sum_partials16(const float* part, long rows, float* out, hipStream_t st);

// ---- transposed convolution support
// D[img][Hs][Ws][16] = dz[img][y/s][x/s][:] where y, x are multiples of s inside the Ho x Wo map, else 0
void launch_zero_stuff(const float* dz, int nimg, int Ho, int Wo, int stride, int Hs, int Ws, float* D, hipStream_t st);
void launch_add_inplace(float* a, const float* b, long n, hipStream_t st);

// ---- weight gradient of a thin convolution: x [nimg][H][W][Cpix] (Cin real channels), dz [nimg][Ho][Wo][16]
// dW[n][c][ky][kx] += sum dz[.,oy,ox,n] * x[., oy*s - pad + ky, ox*s - pad + kx, c]   (OIHW, N = 16)
// scratch: at least wgrad_scratch_floats(...) floats
long wgrad_scratch_floats(int nimg, int Ho, int Cin, int KH, int KW);
void launch_conv_wgrad(const float* x, int Cpix, int Cin, int nimg, int H, int W, const float* dz, int Ho, int Wo, int KH,
                       int KW, int stride, int pad, float* scratch, float* dW, hipStream_t st);
// depthwise 1x1 in front of the encoder: x0[c] = xn[c]*w[c] + b[c] with xn = flow/std. Its gradients and the stem
// conv's weight gradient all follow from A = conv_wgrad of the stem taken on the auxiliary input (xn0, xn1, 1)
// (launch_prep_flow_aux, Cin = 3): see stem_combine_kernel. No data gradient of the stem conv is ever formed.
void launch_prep_flow_aux(const float* flow, int nimg, int H, int W, float* out4, hipStream_t st);
void launch_stem_combine(const float* A /*[16][3][taps]*/, const float* W1 /*[16][2][taps]*/, const float* w, const float* b,
                         int taps, float* dW1, float* dw, float* db, hipStream_t st);

// ---- dense algebra on small matrices (row-major): C[M][N] = beta*C + op(A)*op(B) (+ bias[N] broadcast over rows)
void launch_gemm(bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                 int ldc, float beta, const float* bias, hipStream_t st);
void launch_colsum(const float* X, int rows, int cols, int ld, float* out /* += */, hipStream_t st);
void launch_mish_fwd(const float* z, float* a, long n, hipStream_t st);
void launch_mish_bwd(const float* dy, const float* z, float* dz, long n, hipStream_t st);
// [nimg][P][16] <-> [nimg][16][P]
void launch_nhwc_to_chw(const float* src, int nimg, int P, float* dst, hipStream_t st);
void launch_chw_to_nhwc(const float* src, int nimg, int P, float* dst, hipStream_t st);

// ---- LSTM cell (torch gate order i, f, g, o; hidden 512). pre[B][2048] = x W_ih^T + h W_hh^T + b_ih + b_hh
// act[B][2048] (sigmoid/tanh applied), c_out = f*c_in + i*g, tanhc = tanh(c_out), h_out = o*tanhc
void launch_lstm_fwd(const float* pre, const float* c_in, int B, float* act, float* c_out, float* tanhc, float* h_out,
                     hipStream_t st);
// dpre[B][2048], dc_in[B][512] from dh[B][512] (all consumers summed), dc_out[B][512] (from the next step; may be null)
void launch_lstm_bwd(const float* dh, const float* dc_out, const float* act, const float* c_in, const float* tanhc, int B,
                     float* dpre, float* dc_in, hipStream_t st);

// ---- loss (odometry/loss.py, alpha = 1): L = mean_b sum_t (delta*|dt|^2 + khi*|dr|^2); also the gradients
void launch_clvo_loss(const float* pred_rot, const float* pred_tr, const float* true_rot, const float* true_tr, int B, int T,
                      float* loss /*[1]*/, float* d_rot, float* d_tr, hipStream_t st);

// ---- AdamW (torch.optim.AdamW, amsgrad off) on a flat range; t = 1-based step
void launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float wd, float eps, float beta1,
                  float beta2, int t, hipStream_t st);

}  // namespace atdn

namespace atdn {
// ---- 16 -> 16 channel convolution on NHWC16 maps with v_mfma_f32_16x16x4_f32 (exact fp32): the thin convolutions of
// the CLVO encoder fill a 32x32 MFMA tile to a quarter (N = 16, K rows padded 48 -> 64); here N is exactly one
// 16-column tile, K = KH*KW*16 needs no padding, the whole weight tensor sits in operand registers for the lifetime
// of the block and the input is read from an LDS halo patch with one ds_read_b128 per tap and 16-pixel tile.
// w: OIHW [16][16][K][K]; transposed = use w[c][n][K-1-ky][K-1-kx] instead (data gradient of a convolution).
// z[img][oy][ox][n] = bias[n] + sum x[img][oy*S - pad + ky][ox*S - pad + kx][c] * w(n, c, ky, kx)
// the 7x7 stride-2 pad-3 stem (2 -> 16 channels) on NHWC4 input, same MFMA; w: OIHW [16][2][7][7]
// eval-mode tail fused into the store (inference head): BN(Mish(.)) with the folded affine sc/sh, and with `skip`
// ([nimg][Ho][Wo][16]) the ResidualConv tail BN2(Mish(BN1(Mish(.)) + skip))
struct Conv16Tail {
  const float* sc = nullptr; const float* sh = nullptr;
  const float* skip = nullptr; const float* sc2 = nullptr; const float* sh2 = nullptr;
};
// Training forward only: BatchNorm statistics of Mish(z) taken in the producing kernel (see c16_stat_flush in train_kernels.hip).
// `part` [groups][rows][2][16] with `capacity` floats; the launcher zeroes what it uses and sets `rows` (partial rows per group) for
// launch_bn_finalize_rows. group_imgs = images per statistics group (the images of one time step).
struct Conv16Stats {
  float* part = nullptr; int group_imgs = 1; long capacity = 0; int rows = 0;
};
void launch_stem16(const float* x4, int nimg, int H, int W, const float* w, const float* bias, float* z, hipStream_t st,
                   const Conv16Tail* tail = nullptr, Conv16Stats* stat = nullptr);
// data gradient of a stride-2 16 -> 16 convolution (w OIHW, K = 3 pad 1 or K = 1 pad 0): dx [nimg][H][W][16] from
// dz [nimg][Ho][Wo][16]; accumulate: dx += instead of dx =
void launch_tconv16_s2(const float* dz, int nimg, int Ho, int Wo, const float* w, int K, int pad, int H, int W, bool accumulate,
                       float* dx, hipStream_t st);
void launch_conv16(const float* x, int nimg, int H, int W, const float* w, bool transposed, const float* bias, int K, int S,
                   int pad, float* z, hipStream_t st, bool accumulate = false,   // accumulate: z += instead of z =
                   Conv16Stats* stat = nullptr);
// the same convolution with an eval-mode tail: K = 3 with S = 1 or 3 (Conv blocks), K = 3, S = 2 with tail.skip (ResidualConv)
void launch_conv16_eval(const float* x, int nimg, int H, int W, const float* w, const float* bias, int K, int S, int pad,
                        const Conv16Tail& tail, float* z, hipStream_t st);
}  // namespace atdn
