/* libatdn_hip — C ABI of the MI355X-native ATDN vSLAM visual-odometry inference path.
 *
 * The reference has no FFI layer: its boundary is two Python nn.Module call contracts. Each entry point
 * below names the reference interface it stands in for. All tensor arguments are raw DEVICE pointers to
 * dense fp32 buffers (tensor.data_ptr()) unless marked host; `stream` is a hipStream_t
 * (torch.cuda.current_stream().cuda_stream), 0/NULL = the default stream.
 *
 * Every function returns 0 on success and non-zero on error; atdn_last_error() then holds the message
 * (thread-local). Handles are not re-entrant: one handle per (device, thread), like the reference modules.
 */
#ifndef ATDN_HIP_H
#define ATDN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atdn_gma atdn_gma;   /* RAFTGMA flow network handle */
typedef struct atdn_clvo atdn_clvo; /* ATDNVO pose head handle */
typedef struct atdn_vae atdn_vae;   /* MappingVAE encoder handle (relocalisation embedding) */
typedef struct atdn_clvo_trainer atdn_clvo_trainer; /* ATDNVO training-iteration handle */

int atdn_version(void);
const char* atdn_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * GMA optical flow  —  replaces torch.nn.DataParallel(RAFTGMA(GMA_Parameters()))
 *   construction + checkpoint: atdn_vslam/slam_framework/neural_slam.py:51-53
 *   forward:                   whl:GMA/core/network.py:72-129 (called at neural_slam.py:202,395)
 * ------------------------------------------------------------------------------------------------- */

/* H, W: frame size after the caller's resize/pad (multiples of 8; 376x1232 in NeuralSLAM,
 * neural_slam.py:54,198-199). max_batch: largest number of frame pairs per forward.
 * precision: ATDN_PRECISION_F32 = every GEMM on the exact-fp32 matrix core (v_mfma_f32_32x32x2_f32);
 *            ATDN_PRECISION_SPLIT_F16 = every channel-wide GEMM as three f16 MFMAs on split operands
 *            (x = hi + lo, fp32 accumulate): fp32-grade results at 5.3x the matrix rate (the default);
 *            ATDN_PRECISION_F16 = the fast mode: the same kernels and tensors, but only the hi x hi MFMA of every
 *            product (plain f16 operands, fp32 accumulate) — what the reference itself runs on a GPU under
 *            `mixed_precision` autocast (utils/gma_parameters.py:11); flow within ~1e-2 px of the fp32 CPU path. */
#define ATDN_PRECISION_F32 0
#define ATDN_PRECISION_SPLIT_F16 1
#define ATDN_PRECISION_F16 2
int atdn_gma_create(atdn_gma** out, int H, int W, int max_batch, int precision);

/* Low-latency form for the reference's per-frame call pattern — NeuralSLAM.__call__ runs the flow network on ONE pair per
 * frame (neural_slam.py:202; evaluate_odometry.py:63-66 likewise): launches that would leave most of the chip idle at one to
 * four pairs are cut finer (attention x V along its key axis, with fp32 partial sums). Same results within rounding (another
 * summation order: 8e-5 px at KITTI size after 12 iterations), NOT bit-identical to the default path, whose clip / continued / pair modes are bit-identical to
 * each other. Call between atdn_gma_create and atdn_gma_finalize; `on` = 0 / 1. */
int atdn_gma_set_low_latency(atdn_gma* h, int on);

/* One state-dict entry (load_state_dict, neural_slam.py:52). `key` as in the checkpoint, with or without the
 * DataParallel "module." prefix; `data` is a HOST fp32 buffer of the given shape. Non-float buffers
 * (num_batches_tracked, rel_ind) need not be passed. */
int atdn_gma_load(atdn_gma* h, const char* key, const float* data, const int64_t* shape, int rank);

/* Packs weights for the MFMA kernels (BatchNorm folded), uploads them, allocates the HBM workspace. */
int atdn_gma_finalize(atdn_gma* h);

/* flow_low, flow_up = flow_net(image1, image2, iters=iters, flow_init=flow_init, test_mode=True)
 *   im1, im2   [B,3,H,W]   RGB 0..255
 *   flow_init  [B,2,H/8,W/8] or NULL          (network.py:103-104)
 *   flow_low   [B,2,H/8,W/8], flow_up [B,2,H,W]; channel 0 = x flow, 1 = y flow */
int atdn_gma_forward(atdn_gma* h, const float* im1, const float* im2, int B, int iters, const float* flow_init,
                     float* flow_low, float* flow_up, void* stream);

/* flow_predictions = flow_net(image1, image2, iters=iters, flow_init=flow_init, test_mode=False)   (network.py:106-129:
 * the training-time return of RAFTGMA.forward — the convex upsampling of EVERY iteration's flow with that iteration's mask)
 *   flow_predictions  [iters,B,2,H,W]; slice iters-1 equals atdn_gma_forward's flow_up
 * The mask head runs once per iteration in this call (atdn_gma_forward needs only the last one's); launched kernel by kernel,
 * not as a graph. Inference only: no gradients are produced. */
int atdn_gma_forward_predictions(atdn_gma* h, const float* im1, const float* im2, int B, int iters, const float* flow_init,
                                 float* flow_predictions, void* stream);

/* The same for B consecutive pairs of one clip, as NeuralSLAM walks a sequence (neural_slam.py:196-217: frame t is
 * image2 of pair t-1 and image1 of pair t): frames [B+1,3,H,W]; pair b = (frames[b], frames[b+1]). The feature
 * network runs once per frame instead of twice. Split-f16 handles only. */
int atdn_gma_forward_sequence(atdn_gma* h, const float* frames, int B, int iters, const float* flow_init,
                              float* flow_low, float* flow_up, void* stream);
/* The same for the NEXT clip of one sequence on this handle: frames[0] must be the frame that was frames[B] of the
 * previous atdn_gma_forward_sequence(_continued) call; its features are reused (device-side copy) and the feature
 * network runs on frames[1..B] only, so every frame of a long sequence passes through it exactly once. Results equal
 * the non-continued call's (and atdn_gma_forward's) bit for bit: no kernel's summation order depends on how many images
 * share a launch. */
int atdn_gma_forward_sequence_continued(atdn_gma* h, const float* frames, int B, int iters, const float* flow_init,
                                        float* flow_low, float* flow_up, void* stream);

/* Copies an internal activation to a HOST buffer for parity tests ("fmap", "pyr0".."pyr3", "attn", "net",
 * "x", "corrfeat", "cor1" (relu(convc1(lookup)) of the last iteration), "mask", "coords1", "flow4", "qk", "img4"). Returns the number of floats copied, -1 on error.
 * "sf_clamped" returns ONE float: how many values the split-f16 storage format had to clamp (|x| > 65504, or NaN)
 * on this device since the last such read, and resets the count — non-zero means the activations of this checkpoint
 * left the format's range and results are not fp32-grade (use ATDN_PRECISION_F32). */
long atdn_gma_debug_read(atdn_gma* h, const char* name, float* host, long capacity, void* stream);

/* Per-stage device time (ms, summed over `reps` eager forwards of batch B), measured with HIP events on
 * `stream`. ms_out has ATDN_GMA_STAGES entries: fnet, corr, pool, cnet, attention (row softmax), lookup, motion_encoder
 * (without convc1), aggregate (the attention x V kernel alone), gru_zr (fused z|r convolution, horizontal 1x5 pass),
 * gru_q (horizontal pass), flow_head, mask, gru_ctx (once-per-pair context part of the GRU convolutions; split-f16
 * pipeline only), attn_logits (q,k projection + first QK^T sweep), agg_vt (the v^T projection in front of attention x V),
 * convc1 (the 1x1 convolution behind the lookup; zero when it is fused into the lookup), gru_zr_v / gru_q_v (the vertical
 * 5x1 passes).
 * Stages that hold launches of ONE kernel (aggregate, gru_zr, gru_zr_v, gru_q, gru_q_v, lookup, corr) divide into
 * per-launch times. */
#define ATDN_GMA_STAGES 18
int atdn_gma_profile(atdn_gma* h, int B, int iters, int reps, float* ms_out, void* stream);
/* The same for one of the three call forms: mode 0 = atdn_gma_forward (pair mode, 2B feature-network passes; what
 * atdn_gma_profile times), 1 = atdn_gma_forward_sequence (B + 1 passes), 2 = atdn_gma_forward_sequence_continued (B passes:
 * a continued clip of a long sequence, the form bench.py times). Split-f16 / f16 handles only for modes 1 and 2. */
int atdn_gma_profile_mode(atdn_gma* h, int B, int iters, int reps, int mode, float* ms_out, void* stream);

size_t atdn_gma_workspace_bytes(atdn_gma* h);
void atdn_gma_destroy(atdn_gma* h);

/* ---------------------------------------------------------------------------------------------------
 * CLVO pose head  —  replaces ATDNVO()  (atdn_vslam/odometry/network.py:20-162)
 *   construction: evaluate_odometry.py:124, neural_slam.py:57-59 ; forward: network.py:122-146
 * The module's hidden LSTM attributes become an explicit state tensor so the stateless CNN encoder can be
 * sharded over frame pairs while the recurrence runs as one ordered scan.
 * ------------------------------------------------------------------------------------------------- */

/* H, W: flow size (must reduce to a 16x4x13 map: H in [353,448], W in [1217,1312]); else an error like the
 * reference's Linear(832) shape error. */
int atdn_clvo_create(atdn_clvo** out, int H, int W, int max_batch);
int atdn_clvo_load(atdn_clvo* h, const char* key, const float* data, const int64_t* shape, int rank);
int atdn_clvo_finalize(atdn_clvo* h);

/* features = encoder_CNN(normalize_flow(flows))   (network.py:131-134): flow [B,2,H,W] -> feat [B,512] */
int atdn_clvo_encode(atdn_clvo* h, const float* flow, int B, float* feat, void* stream);

/* T ordered recurrent steps (network.py:137-146) for Bs independent sequences:
 *   feat [T,Bs,512]; state [4,Bs,512] = lstm1_h, lstm1_c, lstm2_h, lstm2_c (in/out; zeros == reset_lstm());
 *   rot, tr [T,Bs,3] (Euler yxz radians, translation). */
int atdn_clvo_step(atdn_clvo* h, const float* feat, int T, int Bs, float* state, float* rot, float* tr, void* stream);
void atdn_clvo_destroy(atdn_clvo* h);

/* ---------------------------------------------------------------------------------------------------
 * CLVO training iteration  —  replaces the body of train() in train_odometry.py:21-49 for one batch:
 *   model.train(); T x model(fl[:, j]) with the LSTM state carried; CLVO_Loss(alpha = 1); loss.backward();
 *   optimizer.step() (AdamW, train_odometry.py:99); model.reset_lstm().
 * BatchNorm layers use per-call batch statistics and update their running averages (momentum 0.1), as torch does.
 * Gradients sit in ONE flat device buffer (`atdn_clvo_trainer_gradients`): data-parallel training all-reduces that
 * buffer (RCCL, averaged over ranks) between forward_backward and adamw_step — see atdn_vslam_amd/training.py.
 * State-dict keys as in ATDNVO().state_dict().
 * ------------------------------------------------------------------------------------------------- */
int atdn_clvo_trainer_create(atdn_clvo_trainer** out, int H, int W, int batch, int sequence_length);
int atdn_clvo_trainer_load(atdn_clvo_trainer* h, const char* key, const float* data, const int64_t* shape, int rank);
int atdn_clvo_trainer_finalize(atdn_clvo_trainer* h);
/* flows [batch, T, 2, H, W], true_rot / true_tr [batch, T, 3] (device). Returns the loss; pred_rot / pred_tr
 * [batch, T, 3] (device) are optional. Overwrites the gradient buffer, advances the BatchNorm running statistics. */
int atdn_clvo_trainer_forward_backward(atdn_clvo_trainer* h, const float* flows, const float* true_rot, const float* true_tr,
                                       float* pred_rot, float* pred_tr, float* loss_out, void* stream);
int atdn_clvo_trainer_gradients(atdn_clvo_trainer* h, float** device_ptr, long* count);
/* torch.optim.AdamW (betas 0.9 / 0.999) on every parameter forward() uses; step is 1-based */
int atdn_clvo_trainer_adamw_step(atdn_clvo_trainer* h, float lr, float weight_decay, float eps, int step, void* stream);
/* copy a named tensor to the host: kind 0 parameter, 1 gradient, 2 BatchNorm running statistic; returns the count or -1 */
long atdn_clvo_trainer_read(atdn_clvo_trainer* h, const char* key, int kind, float* host, long capacity, void* stream);
void atdn_clvo_trainer_destroy(atdn_clvo_trainer* h);

/* ---------------------------------------------------------------------------------------------------
 * MappingVAE encoder  —  replaces the embedding half of atdn_vslam/localization/network.py `MappingVAE.forward`
 * (57-70, non-variational: mu = mean_lin(encoder(get_rgb_norm()(image)))), which NeuralSLAM's relocalisation
 * evaluates for every keyframe and query (slam_framework/neural_slam.py:88-103,158-164,355-383). State-dict keys
 * as in MappingVAE().state_dict() (encoder.*, mean_lin.*); decoder.* keys are accepted and ignored by the host
 * mirror (the decoder only feeds the VAE's training loss, which stays on stock PyTorch).
 * ------------------------------------------------------------------------------------------------- */
int atdn_vae_create(atdn_vae** out, int H, int W, int max_batch);
int atdn_vae_load(atdn_vae* h, const char* key, const float* data, const int64_t* shape, int rank);
int atdn_vae_finalize(atdn_vae* h);
/* size of the embedding map: six stride-2 blocks, 376x1232 -> 6x20 */
int atdn_vae_embedding_shape(const atdn_vae* h, int* out_h, int* out_w);
/* images [B,3,H,W] float32 with values 0..255 (device) -> mu [B][out_h*out_w][128] channels-last (device) */
int atdn_vae_encode(atdn_vae* h, const float* images, int B, float* mu, void* stream);
void atdn_vae_destroy(atdn_vae* h);

/* ---------------------------------------------------------------------------------------------------
 * Pose algebra (host, no GPU)  —  replaces atdn_vslam/utils/transforms.py
 * ------------------------------------------------------------------------------------------------- */

/* transform(rot, tr) (transforms.py:97-119, euler2matrix "yxz" :79-81): HOST rot[3], tr[3] -> row-major 4x4 */
int atdn_pose_transform_f32(const float* rot, const float* tr, float* mat16);
/* rel2abs (transforms.py:147-170): HOST rot[T,3], tr[T,3] -> poses [T+1,4,4] float64, identity first */
int atdn_pose_rel2abs(const float* rot, const float* tr, int T, double* poses);
/* NeuralSLAM's running pose (neural_slam.py:204-207): pose(4x4 fp32, in/out) = pose @ transform(rot, tr) */
int atdn_pose_accumulate_f32(float* pose16, const float* rot, const float* tr);

/* ---------------------------------------------------------------------------------------------------
 * Frame front-end  —  replaces, per camera frame, `im.to(device)`, `TF.resize(im, (376, 1232))` and
 * `InputPadder.pad` (neural_slam.py:197-199,219-221; whl:GMA/core/utils/utils.py:8-20)
 * ------------------------------------------------------------------------------------------------- */

/* torchvision's tensor resize is bilinear, align_corners = False; it antialiases from torchvision 0.17 on
 * (F.interpolate(..., antialias=True)) and does not before that. The reference pins no version
 * (/root/reference/pyproject.toml:14-16), so both are served; ANTIALIAS is the default everywhere. */
#define ATDN_RESIZE_BILINEAR 0
#define ATDN_RESIZE_ANTIALIAS 1

/* src [planes,Hin,Win] (planes = any product of leading dims, e.g. B*3) -> dst [planes,Hout,Wout]; antialiased.
 * Weight tables are cached per (device, geometry, mode); no intermediate buffer: safe on any number of streams. */
int atdn_resize_frames(const float* src, int planes, int Hin, int Win, int Hout, int Wout, float* dst, void* stream);
int atdn_resize_frames_mode(const float* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst,
                            void* stream);
/* the same from uint8 pixels already on the device (the conversion to fp32 is fused into the resize) */
int atdn_resize_frames_u8(const uint8_t* src, int planes, int Hin, int Win, int Hout, int Wout, int antialias, float* dst,
                          void* stream);

/* F.pad(x, [left, right, top, bottom], mode="replicate") as InputPadder.pad applies it (utils.py:19-20):
 * src [planes,H,W] -> dst [planes,H+top+bottom,W+left+right] */
int atdn_pad_frames(const float* src, int planes, int H, int W, int left, int right, int top, int bottom, float* dst,
                    void* stream);

/* uint8 camera frames in HOST memory -> fp32 frames at the network size on the device, in one call:
 * asynchronous H2D copy (pinned host memory; pageable memory works but serialises) on the handle's own copy stream
 * into one of two device staging slots, then resize + uint8->fp32 on `stream`. Consecutive calls alternate slots, so
 * the copy of the next clip overlaps the flow network of the current one.
 * LIFETIME: the copy is asynchronous. `host_frames` must stay valid (and unmodified) until the copy has completed:
 * that is guaranteed once the SECOND-NEXT call of atdn_ingest_frames_u8 on the same handle has returned (a call waits
 * on the host for the copy that last used its staging slot), or once `stream` has been synchronised after this call.
 * A caller that keeps the host buffers of its last two calls alive is safe (pipeline.FrameIngest does).
 *   host_frames [n_frames,3,Hin,Win] uint8 (HOST) -> dst [n_frames,3,Hout,Wout] fp32 (DEVICE), n_frames <= max_frames */
typedef struct atdn_ingest atdn_ingest;
int atdn_ingest_create(atdn_ingest** out, int Hin, int Win, int Hout, int Wout, int max_frames, int antialias);
int atdn_ingest_frames_u8(atdn_ingest* h, const uint8_t* host_frames, int n_frames, float* dst, void* stream);
void atdn_ingest_destroy(atdn_ingest* h);

/* ---------------------------------------------------------------------------------------------------
 * Individual kernels, exported for unit parity tests and roofline micro-benchmarks
 * ------------------------------------------------------------------------------------------------- */

/* CorrBlock.__call__ (whl:GMA/core/corr.py:32-53): pyramid level l is [B*H8*W8][H_l*W_l] with
 * H_l = H8 >> l, W_l = W8 >> l; coords [B*H8*W8][2] (x,y); out [B*H8*W8][ldo], channel l*81 + i*9 + j. */
int atdn_corr_lookup(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3, int B, int H8,
                     int W8, const float* coords, float* out, int ldo, void* stream);

/* CorrBlock.__init__ (corr.py:16-30,55-63): fmap1, fmap2 channels-last [B][H8*W8][C] (C % 32 == 0) ->
 * pyr0..pyr3 as above. */
int atdn_corr_pyramid(const float* fmap1, const float* fmap2, int B, int H8, int W8, int C, float* pyr0, float* pyr1,
                      float* pyr2, float* pyr3, void* stream);

/* The same two operations through the kernels of the DEFAULT (split-f16) path — what atdn_gma_forward launches and
 * bench.py times: corr_bricks_kernel for all four levels (corr.py:16-30,55-63; levels 1-3 from 2x2-pooled target features,
 * avg_pool2d commutes with the dot product), the brick-major pyramid, and lookup_conv_kernel (corr.py:32-53 +
 * utils/utils.py:59-73 bilinear_sampler; fused with convc1 of update.py:76-78). fmap1, fmap2 fp32 channels-last
 * [B][H8*W8][256] (C must be 256); coords [B*H8*W8][2] (x, y) or NULL when no lookup is wanted. Every output is optional
 * (NULL skips it): pyr0..pyr3 = the levels un-bricked to the reference's row-major [B*H8*W8][H_l*W_l]; samples
 * [B*H8*W8][324], channel l*81 + i*9 + j (the FUSED = false instantiation of the sampling code); cor1 [B*H8*W8][256] =
 * relu(convc1(samples)) from the fused instantiation, with convc1's HOST weight [256][324] (torch layout [256,324,1,1])
 * and HOST bias [256]. Synchronises the stream before returning. */
int atdn_corr_lookup_bricks(const float* fmap1, const float* fmap2, int B, int H8, int W8, int C, const float* coords,
                            float* pyr0, float* pyr1, float* pyr2, float* pyr3, float* samples,
                            const float* convc1_weight_host, const float* convc1_bias_host, float* cor1, void* stream);

/* Generic NHWC convolution through the implicit-GEMM MFMA engine: src [nimg][H][W][Cin] (Cin % 32 == 0 or
 * Cin in {4,16}), HOST weight in torch layout [Cout][Cin][KH][KW] (+ HOST bias or NULL), dst
 * [nimg][Ho][Wo][Cout]; relu != 0 applies ReLU. */
int atdn_conv2d_nhwc(const float* src, int nimg, int H, int W, int Cin, const float* weight_host,
                     const float* bias_host, int Cout, int KH, int KW, int stride, int padH, int padW, int relu,
                     float* dst, void* stream);

/* The same convolution through the split-f16 engine (three f16 MFMAs per product, fp32-grade result):
 * src fp32 NHWC (converted internally to the sf format), Cin % 32 == 0, fp32 NHWC output. */
int atdn_conv2d_nhwc_sf(const float* src, int nimg, int H, int W, int Cin, const float* weight_host,
                        const float* bias_host, int Cout, int KH, int KW, int stride, int padH, int padW, float* dst,
                        void* stream);
/* ... with the epilogue named: sf_store = 0 writes fp32 (what atdn_conv2d_nhwc_sf does), sf_store = 1 writes the split-f16
 * format through the channel-vector store the product's layers use (Cout % 32 == 0) and decodes it to fp32 `dst`. */
int atdn_conv2d_nhwc_sf_epi(const float* src, int nimg, int H, int W, int Cin, const float* weight_host,
                            const float* bias_host, int Cout, int KH, int KW, int stride, int padH, int padW, int sf_store,
                            float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATDN_HIP_H */
