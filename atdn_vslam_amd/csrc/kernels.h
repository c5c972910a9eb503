// Non-GEMM kernels of the odometry path (host launchers; implementations in kernels.hip).
#pragma once
#include "common.h"

namespace atdn {

bool& sf_fast_mode();  // see conv_sf.h

struct PyramidLevels {  // correlation pyramid of ONE batch: level l is [B*N][H_l*W_l] fp32
  const float* base[4];
  int H[4], W[4];
};

// frames NCHW [B,3,H,W] (0..255) x2 -> NHWC4 [2B][H][W][4] = 2*(x/255)-1, 4th channel 0   (network.py:75-76)
// n2 = number of images taken from im2 (B for pair mode; 1 in sequence mode: only the clip's last frame)
void launch_prep_images(const float* im1, const float* im2, int B, int H, int W, float* img4, hipStream_t st, int n2);

// 7x7 / stride-2 stem of the GMA encoders on the split-f16 engine (stem_sf.hip). img4: NHWC4 fp32 frames; wfrag / wscale:
// pack_stem_sf copy (weights.h). mode 0: relu(conv + bias) -> out_sf [nimg][Ho*Wo][64]; mode 1: InstanceNorm partials only
// (part_* as launch_in_finalize_cnt reads them, stem_sf_groups(Ho, Wo) groups per image); mode 2: relu((conv + bias -
// mean) * rstd) -> out_sf with mean / rstd [nimg][64]
int stem_sf_groups(int Ho, int Wo);
void launch_stem_sf(int mode, const float* img4, int nimg, int H, int W, const float* wfrag, float wscale,
                    const float* bias, float* out_sf, float* part_sum, float* part_m2, float* part_cnt,
                    const float* mean, const float* rstd, hipStream_t st);

// InstanceNorm statistics from the conv epilogue's per-(32-row group) partials -> mean, rstd [nimg][C], merged in fp64
// in two levels (32 slabs of groups per image, then one merge in a fixed order).
// part_cnt [nimg][groups] = valid rows per group as the 2-D tiled conv kernels report them; nullptr for the 1-D tiled
// kernels (rows follow from the group index).
void launch_in_finalize_cnt(const float* part_sum, const float* part_m2, const float* part_cnt, int nimg,
                            int groups_per_img, int HW, int C, float eps, float* mean, float* rstd, double* scratch,
                            hipStream_t st);  // scratch: nimg * 32 * C * 4 doubles
// y = relu((x-mean)*rstd); optional residual: y = relu(r + y), r = res or (res-rmean)*rrstd when rmean given
void launch_in_apply(float* x, const float* mean, const float* rstd, const float* res, const float* rmean,
                     const float* rrstd, int nimg, long HW, int C, hipStream_t st);

// 2x2/stride-2 average pooling of the last two dims of [rows][H][W] (floor)   (corr.py:28-30)
void launch_avgpool(const float* src, int H, int W, float* dst, long rows, hipStream_t st);

// radius-4 bilinear pyramid lookup  (corr.py:32-53): out[p][l*81 + i*9 + j], ldo >= 324
void launch_lookup(const PyramidLevels& pyr, const float* coords1, long npix_total, float* out, int ldo,
                   hipStream_t st);

// in-place row softmax of [rows][ld] over the first n columns; columns [n, ld) are zeroed   (gma.py:74)
void launch_softmax_rows(float* x, long rows, int n, int ld, hipStream_t st);

// coords1 = grid (+ flow_init NCHW [B,2,H8,W8]); flow4 / x flow channels = coords1 - coords0
void launch_init_coords(const float* flow_init, int B, int H8, int W8, float* coords1, float* flow4, float* xflow,
                        int ldx, hipStream_t st);

// convex 8x upsampling (network.py:59-70) straight into caller tensors: flow_up NCHW [B,2,8H8,8W8], flow_low [B,2,H8,W8]
void launch_upsample(const float* mask, const float* flow4, int B, int H8, int W8, float* flow_low, float* flow_up,
                     hipStream_t st);

// ---- CLVO head
// flow NCHW [B,2,H,W] -> NHWC4: (x/std_c)*dw_w[c] + dw_b[c]    (normalizations.py:8-10, odometry/network.py:64)
void launch_prep_flow(const float* flow, int B, int H, int W, const float* dw_w, const float* dw_b, float* out4,
                      hipStream_t st);
// MappingVAE input normalisation: images NCHW [B,3,H,W] (0..255) -> NHWC4 (x/255 - mean)/std, 4th channel 0
void launch_prep_rgb(const float* images, int B, int H, int W, float* out4, hipStream_t st);
// y[b][n] = act( W0[n]·x0[b] (+ W1[n]·x1[b]) + b0[n] (+ b1[n]) ),  act: 0 none, 1 mish
void launch_linear(const float* W0, const float* x0, int K0, int ldx0, const float* W1, const float* x1, int K1,
                   int ldx1, const float* b0, const float* b1, int act, float* y, int ldy, int N, int B,
                   hipStream_t st);
// One time step of the recurrent tail as a three-stage pipeline (see lstm_pipe_kernel): stage 1 = lstm1 cell on
// pre1 (= W_ih1 x + b_ih1, batched beforehand) and h1_in; stage 2 = lin_out = Mish(Wlin lin_in + blin); stage 3 = lstm2
// cell on x2_in and h2_in with its input projection inline. Stages with do_* == 0 are skipped (pipeline fill / drain).
struct LstmPipeArgs {
  int Hd, B, do1, do_lin, do2;
  const float *pre1, *Whh1, *bhh1, *h1_in; float *c1, *h1_out;
  const float *Wlin, *blin, *lin_in; float* lin_out;
  const float *Wih2, *bih2, *Whh2, *bhh2, *x2_in, *h2_in; float *c2, *h2_out;
};
void launch_lstm_pipe(const LstmPipeArgs& a, hipStream_t st);
// The same recurrence for ONE batch row as one persistent launch per sequence (lstm_scan.hip): weights resident in registers,
// the three pipeline stages exchange their 512-value results through tagged 8-byte granules. `state` [4][512] = h1, c1, h2, c2
// (read at the start, written at the end), `h2seq` [T][512] = lstm2's hidden state after every step, `exchange` = a device
// buffer of lstm_scan_exchange_bytes() that this call zeroes on the stream before the launch.
long lstm_scan_exchange_bytes();
bool lstm_scan_fits_device();   // all 128 workgroups resident at once on the current device?
void launch_lstm_scan(const float* pre1, const float* Whh1, const float* bhh1, const float* Wlin, const float* blin,
                      const float* Wih2, const float* bih2, const float* Whh2, const float* bhh2, float* state, float* h2seq,
                      void* exchange, int T, hipStream_t st);
// both regressors (512 -> 128 -> 64 -> 3, Mish, last layer no bias): out rot [B][3], tr [B][3]
struct MlpHead { const float *w0, *b0, *w1, *b1, *w2; };
void launch_mlp_heads(const float* h2, int B, MlpHead rot, MlpHead tr, float* rot_out, float* tr_out, hipStream_t st);

// ---- split-f16 ("sf", sf.h) variants used by the 3xf16 MFMA pipeline
// fp32 NHWC [rows][C] (C % 32 == 0) -> sf
void launch_to_sf(const float* src, float* dst, long rows, int C, hipStream_t st);
// sf -> fp32 NHWC
void launch_from_sf(const float* src, float* dst, long rows, int C, hipStream_t st);
// InstanceNorm apply, out of place: raw fp32 x -> sf y = relu((x-mean)*rstd); optional residual (sf `res`, or raw fp32
// `res_raw` normalised with rmean/rrstd): y = relu(r + y)
void launch_in_apply_sf(const float* x, float* y, const float* mean, const float* rstd, const float* res,
                        const float* res_raw, const float* rmean, const float* rrstd, int nimg, long HW, int C,
                        hipStream_t st, bool res_relu = false);
// 2x2 average (floor sizes) of an sf feature map [img][H*W][C] (per-image strides sb / db in floats)
void launch_pool_features_sf(const float* src, int nimg, int H, int W, int C, long sb, float* dst, long db, hipStream_t st);
// as launch_init_coords, x flow channels written in sf at channels cflow, cflow+1 of the sf GRU input
void launch_init_coords_sf(const float* flow_init, int B, int H8, int W8, float* coords1, float* flow4, float* x,
                           int ldx, int cflow, hipStream_t st);

void launch_fill(float* p, long n, float v, hipStream_t st);

// ---- bricked correlation pyramid + lookup fused with convc1 (lookup_fused.hip; split-f16 pipeline)
// level l of ONE batch in BRICKS of 4 rows x 8 columns of target cells (32 floats = one 128-byte line). Source pixels are
// grouped in PIXEL BLOCKS of 64 consecutive pixels of a pair (NPB blocks per pair, padded to an even count so that a pair's region
// covers whole 128-pixel strips); inside a pixel block the layout is brick-major:
//   base[l][(((pair * NPB + (p >> 6)) * NBK_l + brick) * 64 + (p & 63)) * 32 + (y & 3) * 8 + (x & 7)],  brick = (y >> 2) * BW_l + (x >> 3),
// NBK_l = BH_l * BW_l = NB[l] / 32; cells past H / W are zeros. (Round 4. The correlation kernel stores whole lines in 4 KB runs —
// 32 pixels x one brick per wave and tile; a lookup block = one pixel block finds everything it reads inside NBK_l x 8 KB, and
// its neighbouring pixels, whose windows share bricks, read neighbouring lines. Rounds 2-3 kept a pixel's whole map contiguous.)
constexpr int kBrickPixelBlock = 64;
struct BrickPyramid {
  const float* base[4];
  int H[4], W[4], BW[4], BH[4], NB[4];
  int N;     // source pixels per pair
  int NPB;   // pixel blocks per pair: 2 * ceil(N / 128)
};
inline int brick_pixel_blocks(int N) { return 2 * ((N + 127) / 128); }
// feature rows [img][H*W][C] -> brick order [img][BH*BW*32][C] (zero rows for padding cells); sb / db per-image strides
void launch_brick_rows(const float* src, long sb, int nimg, int H, int W, int C, float* dst, long db, hipStream_t st);
// one pyramid level in the layout above: out[...] = scale * <f1[pair][p], f2b[pair][n']> for brick n' / 32, cell n' % 32
// (corr_bricks.hip); f1 sf [B][N][256] (per-pair stride sb1 floats), f2b sf [B][NB][256] in brick order (zero rows for padding
// cells), out fp32 with brick_pixel_blocks(N) * 64 * NB floats per pair (rows N .. 64 * NPB - 1 of a pair are scratch)
void launch_corr_bricks(const float* f1, long sb1, const float* f2b, long sb2, int B, int N, int NB, float scale, float* out,
                        bool fast, hipStream_t st);
// bricked level (pairs of N source pixels) -> row-major [npix][H*W] (debug reads)
void launch_unbrick(const float* src, long NB, int N, int H, int W, long npix, float* dst, hipStream_t st);
// cor1 = relu(convc1(lookup(coords1))) (corr.py:32-53 + update.py:76-78): out sf [npix][256]; wfrag = convc1 weights in
// fragment-major order for v_mfma_f32_16x16x32_f16 (weights.h: pack_fragment_major16; K = 352), bias [256]; coords_used (optional) receives the coordinates that were sampled
void launch_lookup_conv(const BrickPyramid& pyr, const float* coords1, long npix, float* coords_used, const float* wfrag,
                        float wscale, const float* bias, float* out, bool fast, hipStream_t st);
// the lookup alone: out sf [npix][352]
void launch_lookup_bricks(const BrickPyramid& pyr, const float* coords1, long npix, float* out, hipStream_t st);

}  // namespace atdn
