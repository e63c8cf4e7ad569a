/*
 * s2l_hip.h -- C-ABI of libs2l_hip.so: the MI355X (gfx950) lip-render hot path of Speech2Lip.
 *
 * The reference (CVMI-Lab/Speech2Lip) has no FFI layer: its boundary for this path is the Python
 * nn.Module surface of `TalkingFace` (src/face_simple/models/tf_nerf.py).  Each entry point below
 * replaces the ATen op sequence behind one reference method; `speech2lip_amd/talking_face.py`
 * binds them with ctypes (INTEGRATION.md shows the stub) and re-provides the module surface.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative S2L_E_* for argument errors, a positive hipError_t otherwise;
 *   - every pointer is DEVICE memory unless the name ends in `_host`; all tensors are dense fp32,
 *     row-major, 16-byte aligned; the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); calls are
 *     asynchronous on that stream and re-entrant across streams.
 */
#ifndef S2L_HIP_H
#define S2L_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* s2l_stream_t;

enum {
  S2L_OK = 0,
  S2L_E_NULL = -1,     /* a required pointer is NULL            */
  S2L_E_SIZE = -2,     /* a size / count argument is invalid    */
  S2L_E_ALIGN = -3,    /* a pointer is not 16-byte aligned      */
  S2L_E_UNSUPPORTED = -5,   /* a kernel form that only libs2l_hip_ref.so holds (the product library keeps one kernel per job) */
  S2L_E_GEOMETRY = -4  /* a box the reference cannot evaluate either: composite lip box ENTIRELY outside the face frame (F.pad raises; a partly-outside box is cropped as F.pad crops), crop/U-Net windows, LPIPS minimum size */
};

/* Index of each state-dict tensor in the pointer table handed to s2l_pack_weights.  Names are
 * the reference's state-dict keys (tf_nerf.py:91-109, :144, :149-172); torch layouts
 * ([out,in] for Linear, [out,in,k] for Conv1d). */
enum {
  S2L_T_CONV0_W = 0, S2L_T_CONV0_B, S2L_T_CONV2_W, S2L_T_CONV2_B, S2L_T_CONV4_W, S2L_T_CONV4_B,
  S2L_T_CONV6_W, S2L_T_CONV6_B, S2L_T_FC1_0_W, S2L_T_FC1_0_B, S2L_T_FC1_2_W, S2L_T_FC1_2_B,
  S2L_T_FC_UV_W, S2L_T_FC_UV_B, S2L_T_FC_AUDIO_W, S2L_T_FC_AUDIO_B, S2L_T_FC_TIME_W, S2L_T_FC_TIME_B,
  S2L_T_FC_UV_SKIP_W, S2L_T_FC_UV_SKIP_B, S2L_T_FC_AUDIO_SKIP_W, S2L_T_FC_AUDIO_SKIP_B,
  S2L_T_FC_TIME_SKIP_W, S2L_T_FC_TIME_SKIP_B,
  S2L_T_PTS0_W, S2L_T_PTS0_B, S2L_T_PTS1_W, S2L_T_PTS1_B, S2L_T_PTS2_W, S2L_T_PTS2_B,
  S2L_T_PTS3_W, S2L_T_PTS3_B, S2L_T_PTS4_W, S2L_T_PTS4_B, S2L_T_PTS5_W, S2L_T_PTS5_B,
  S2L_T_PTS6_W, S2L_T_PTS6_B, S2L_T_PTS7_W, S2L_T_PTS7_B, S2L_T_OUT_W, S2L_T_OUT_B,
  S2L_NUM_TENSORS
};

/* Pad modes of the paste step (tf_nerf.py:345-350 keys them off substrings of cfg.data.path). */
enum { S2L_PAD_MAY = 0, S2L_PAD_DEFAULT = 1 };
/* relative pose of s2l_rel_pose: OBS2CAN = Tc.inv(T) (utils.py:54-58, face_tracker.py:583-584: the pose behind
 * coords/%05d.npy); CAN2OBS = T.inv(Tc) (utils.py:60-71, training.py:263-268); CAN2OBS_INV = inv(T.inv(Tc))
 * (utils.py:73-77, training.py:270-275), which is Tc.inv(T) again and is computed as such. */
enum { S2L_POSE_OBS2CAN = 0, S2L_POSE_CAN2OBS = 1, S2L_POSE_CAN2OBS_INV = 2 };
enum { S2L_SAMPLE_ZEROS = 0, S2L_SAMPLE_BORDER = 1 };

/* Library / build identification: "s2l_hip <version> gfx950". */
const char* s2l_version(void);

/* Number of floats in the packed weight blob. */
int64_t s2l_packed_floats(void);

/* Pack the 42 hot-path state-dict tensors into the device blob the kernels read (MFMA operand
 * order for the 256x256 layers, transposed copies for the small per-frame/per-pixel products,
 * folded first-layer/skip matrices for s2l_rgb_forward).  Re-run after any weight update.
 *   tensors_host : HOST array of S2L_NUM_TENSORS DEVICE pointers, indexed by S2L_T_*
 *   div_term_host: HOST array of 10 floats, PositionalEncodingTime.div_term (tf_nerf.py:431-432)
 *   packed       : device, s2l_packed_floats() floats
 * Replaces: nothing in the reference (it keeps nn.Parameters); this is load-time layout work. */
int s2l_pack_weights(const float* const* tensors_host, const float* div_term_host, float* packed,
                     s2l_stream_t stream);

/* Audio encoder.  windows [B,16,29] -> feat [B,64].
 * Replaces TalkingFace.audio_merge_forward (tf_nerf.py:197-213; layers :91-109). */
int s2l_audio_encode(const float* packed, const float* windows, float* feat, int64_t n_windows,
                     s2l_stream_t stream);

/* One frame per call (fewer than four frames): encoder + frame vectors in ONE launch -- windows [n,16,29] and frame indices [n] -> q0 / q5
 * [n,256] for s2l_render_lip and, if feat != NULL, the features [n,64]; the bits of s2l_audio_encode followed by s2l_frame_vectors
 * (tf_nerf.py:197-213, :247-281; inference.py:129-159 calls the model once per frame).  n >= 4: S2L_E_SIZE. */
int s2l_frame_front(const float* packed, const float* windows, const int64_t* frame_idx, float* feat, float* q0, float* q5, int64_t n,
                    s2l_stream_t stream);

/* Per-frame (pixel-invariant) halves of the first and skip layers for the batched renderer:
 *   q0[f] = W0 (Wa a_f + Wt PE(idx_f) + b_uv + b_a + b_t) + b0
 *   q5[f] = W5[:, :256] (Wa' a_f + Wt' PE(idx_f) + b_uv' + b_a' + b_t') + b5
 * feat [F,64], frame_idx [F] int64 -> q0, q5 [F,256].
 * Replaces the frame-only terms of rgb_forward (tf_nerf.py:247, :252-258, :269-281) including
 * PositionalEncodingTime.__call__ (:434-442).  SURVEY.md §3.3. */
int s2l_frame_vectors(const float* packed, const float* feat, const int64_t* frame_idx, float* q0,
                      float* q5, int64_t n_frames, s2l_stream_t stream);

/* Per-pixel (frame-invariant) halves:  p0[p] = W0 Wuv E(uv_p),  p5[p] = W5[:, :256] Wuv' E(uv_p).
 * coords [HW,2] (u,v) -> p0, p5: opaque tables for s2l_render_lip, each ceil(HW/16)*16*256 floats
 * (16-pixel groups in the renderer's LDS-DMA order).  Replaces Embedder.__call__
 * (tf_nerf.py:404-425) and the fc_uv / fc_uv_skip terms (:252, :269) for a fixed pixel grid
 * (rendering.py:9-28). */
int s2l_pixel_tables(const float* packed, const float* coords, float* p0, float* p5, int64_t hw,
                     s2l_stream_t stream);

/* Fused render of a clip: for every frame f and pixel p
 *   h0 = relu(p0[p] + q0[f]); h1..h4 = relu(W h + b); h5 = relu(p5[p] + q5[f] + W5[:,256:] h4);
 *   h6, h7; rgb = Wout h7 + bout          (no output activation, tf_nerf.py:283)
 * out [F,HW,3]; p0/p5 from s2l_pixel_tables, q0/q5 [F,256] from s2l_frame_vectors.  One persistent
 * launch; activations never leave registers; fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * Replaces the per-frame driver inference.py:140-159 + rgb_forward (tf_nerf.py:225-285). */
int s2l_render_lip(const float* packed, const float* p0, const float* p5, const float* q0,
                   const float* q5, float* out, int64_t hw, int64_t n_frames, s2l_stream_t stream);
/* OPT-IN split-half speed mode of s2l_render_lip (same arguments + packed16; same tile shapes, ring and tables; csrc/render16.hip):
 * the seven 256x256 layers and the output layer on v_mfma_f32_16x16x32_f16 with every fp32 operand x carried as two IEEE HALVES,
 * hi = f16(x), lo = f16(x - hi) (11 + 11 significant bits; the activations' parts by v_cvt_pkrtz_f16_f32, the weights' to nearest),
 * and a product evaluated as W_lo a_hi + W_hi a_lo + W_hi a_hi, fp32 accumulation; the first layer and the skip terms stay exact fp32
 * table sums.  RANGE CONDITION the exact kernel does not have: every weight and every pre-activation must satisfy |x| < 65504 (the half
 * range); beyond it the parts saturate -- the result is wrong but finite, never inf / NaN -- and values below 2^-14 keep only the
 * subnormal halves' absolute precision (~6e-8).  Measured RMSE 1.4e-6 / 117 dB against the CPU oracle at 96x96 (north-star bar: RMSE
 * <= 1e-4), 3.3 - 3.5 x the exact kernel's frame rate.  The exact fp32 kernel is the default everywhere; this one runs only when a caller
 * asks for it (TalkingFace.render_clip(precision="split")).  Replaces the same reference lines as s2l_render_lip
 * (inference.py:140-159, tf_nerf.py:225-285).
 * s2l_pack_render16: the (hi | lo) half A-operand slabs, s2l_render16_packed_halves() uint16 values, from the fp32 blob. */
int64_t s2l_render16_packed_halves(void);
int s2l_pack_render16(const float* packed, void* packed16, s2l_stream_t stream);
int s2l_render_lip_split(const float* packed, const void* packed16, const float* p0, const float* p5, const float* q0,
                         const float* q5, float* out, int64_t hw, int64_t n_frames, s2l_stream_t stream);
/* The renderer has four tile shapes: 16 pixels x 12 frames (clips), 192 pixels x 1 frame (clip lengths that are not multiples of 12),
 * 64 pixels x 1 frame (4 waves x 16 samples, every wave all 256 features) and the FEATURE-SPLIT tile of 16 pixels x 1 frame whose four
 * waves own 64 features each and exchange the activations through LDS (one or a few frames per call, the reference's own mode,
 * inference.py:129,140-159: a 64 x 64 frame takes 66 us instead of 140).  s2l_render_lip picks the one with the smallest estimated time;
 * frames are bit-identical whichever shape rendered them.  s2l_set_render_shape: 0 = choose per call (default), 1 / 2 / 3 / 4 = always
 * the 12-frame / 192-pixel / 64-pixel / feature-split shape (tests and A/B measurements; process-wide). */
int s2l_set_render_shape(int mode);
/* Cap on the persistent renderer's workgroups ON THE CURRENT DEVICE (hipGetDevice of the calling thread; 0 = one per CU, the
 * default).  A multi-GPU host that overlaps RCCL with rendering passes CUs - k so that k CUs stay free for RCCL's kernels;
 * results do not depend on it.  Per-device state held in atomics: safe with one host thread per GPU.  All other one-time
 * per-device launch setup inside the library (CU counts, dynamic-LDS opt-ins) is atomic-guarded the same way. */
int s2l_set_render_cus(int n_workgroups);

/* Exact drop-in for TalkingFace.rgb_forward on arbitrary rows (tf_nerf.py:225-285, May flags):
 * uv_audio [N,66] = (u, v, 64 audio features) per row, one frame index for the call -> out [N,3].
 * xbuf is caller-provided scratch of N*128 floats (embedded rows).  Used by the training-time
 * 4-tap ensemble (training.py:158-251) where coordinates move every step. */
int s2l_rgb_forward(const float* packed, const float* uv_audio, int64_t time_index, float* xbuf,
                    float* out, int64_t n_rows, s2l_stream_t stream);

/* Which kernel runs the general-row forward (s2l_rgb_forward, the ensemble's forward) when no activations are saved: 0 = chosen per call (default:
 * up to three rounds of 16-row tiles whose four waves split the 256 features -- one frame of the reference's per-frame driver,
 * inference.py:152-159, is 4 096 or 9 216 rows -- else 64- / 128-row tiles with a wave per 16-row column), 1 = always the column form, 2 = always the
 * feature-split tile.  Every output is the same chain of MFMAs on the same operands in both: the same bits (a test and A/B aid). */
int s2l_set_rows_kernel(int kind);


/* The embedding half of s2l_rgb_forward on its own: x [N,128] = [E(uv) 42 | audio 64 | PE(time_index) 20 | 0 0] per row
 * (Embedder.__call__ tf_nerf.py:404-425 on the uv columns, PositionalEncodingTime :434-442), for callers that run
 * s2l_train_forward / _backward on it (the autograd of rgb_forward). */
int s2l_embed_rows(const float* packed, const float* uv_audio, int64_t time_index, float* x, int64_t n_rows,
                   s2l_stream_t stream);

/* 4-tap local-ensemble forward of the training step for one frame
 * (Trainer.predict_lip_image, src/face_simple/training.py:158-251): the MLP at
 * clamp(coords + (vx*0.5/W + eps, vy*0.5/H + eps), 0, 1), vx,vy in {-1,1}, eps = (0.5/H)*u01/2,
 * area-weighted with the diagonal swap of :240-245.  coords [N,2]; feat [64] = the frame's audio
 * feature (s2l_audio_encode); u01 = the U(0,1) draw of :200; work = scratch of
 * s2l_predict_lip_image_work_floats(N) floats; out [N,3]. */
int64_t s2l_predict_lip_image_work_floats(int64_t n_pixels);
int s2l_predict_lip_image(const float* packed, const float* coords, const float* feat,
                          int64_t time_index, int width, int height, float u01, float* work,
                          float* out, int64_t n_pixels, s2l_stream_t stream);

/* ---- training step (BASELINE config 5; fp32 exact-parity mode) -------------------------------
 * The reference gets these from torch autograd (training.py:559 `loss.backward()`); the entry
 * points below are the forward-with-saved-activations and the hand-written backward of the same
 * functions.  Row order is caller-defined; x rows are [E(uv) 42 | audio 64 | PE(t) 20 | 0 0]. */

/* Halves of s2l_predict_lip_image: x [4N,128] rows + areas [4N] for one frame; area-weighted
 * reduce pred [4N,3] -> out [N,3] (training.py:204-249). */
int s2l_ensemble_rows(const float* packed, const float* coords, const float* feat, int64_t time_index,
                      int width, int height, float u01, float* x, float* areas, int64_t n_pixels,
                      s2l_stream_t stream);
int s2l_ensemble_reduce(const float* pred, const float* areas, float* out, int64_t n_pixels,
                        s2l_stream_t stream);
/* s2l_ensemble_rows for a whole batch of frames in one launch: feat [F,64], time_index int64 [F] and u01 fp32 [F] on the
 * device; frame f owns rows [4 f HW, 4 (f+1) HW) of x [F*4*HW,128] and areas [F*4*HW]. */
int s2l_ensemble_rows_batch(const float* packed, const float* coords, const float* feat, const int64_t* time_index,
                            const float* u01, int width, int height, float* x, float* areas, int64_t n_pixels,
                            int64_t n_frames, s2l_stream_t stream);
/* d pred [N,3] -> d rgb of the four taps [4N,3]. */
int s2l_ensemble_backward(const float* dpred, const float* areas, float* drgb, int64_t n_pixels,
                          s2l_stream_t stream);
/* The same two kernels over a whole batch of frames in one launch: frame f owns rows [4 f HW, 4 (f+1) HW) of the per-tap
 * arrays (pred rows / drgb [F*4*HW,3], areas [F*4*HW]) and [f HW, (f+1) HW) of the per-pixel ones. */
int s2l_ensemble_reduce_batch(const float* pred, const float* areas, float* out, int64_t n_pixels, int64_t n_frames,
                              s2l_stream_t stream);
int s2l_ensemble_backward_batch(const float* dpred, const float* areas, float* drgb, int64_t n_pixels,
                                int64_t n_frames, s2l_stream_t stream);
/* MLP on rows x [N,128] -> rgb [N,3], saving the post-ReLU activations h0..h7 in hsave [8,N,256]
 * (tf_nerf.py:252-283 with the first/skip projections folded at pack time). */
int s2l_train_forward(const float* packed, const float* x, float* hsave, float* rgb, int64_t n_rows,
                      s2l_stream_t stream);
/* Backward chain: drgb [N,3] -> dzsave [8,N,256] (gradient w.r.t. the pre-activation of h0..h7) and
 * dxa [N,64] (gradient w.r.t. the audio columns of x). */
int s2l_train_backward(const float* packed, const float* drgb, const float* hsave, float* dzsave,
                       float* dxa, int64_t n_rows, s2l_stream_t stream);
/* Scratch floats for the split reductions below, for a result of n_elems elements. */
int64_t s2l_split_work_floats(int64_t n_elems);
/* dw [256,k_in] = dz[:, :256]^T in[:, :k_in] over n_rows rows (k_in = 128 or 256; ld* = row strides
 * in floats) and, when db != NULL, db [256] = column sums of dz (the bias gradient, free in the same
 * pass); fp32 MFMA, deterministic two-stage reduction; work: s2l_split_work_floats(256*k_in). */
int s2l_wgrad(const float* dz, int ldz, const float* in, int ldin, int k_in, float* work, float* dw,
              float* db, int64_t n_rows, s2l_stream_t stream);
/* out [m,c] = a[:, :m]^T b[:, :c] (m <= 4, c <= 256); a == NULL with m == 1: column sums of b
 * (bias gradients).  work: s2l_split_work_floats(m*c). */
int s2l_small_outer(const float* a, int lda, int m, const float* b, int ldb, int c, float* work,
                    float* out, int64_t n_rows, s2l_stream_t stream);
/* out [S,c] = column sums of each segment of rows_per_segment consecutive rows of src [S*rows_per_segment, ld] (c <= 256
 * dividing 256): the per-frame gradient of the audio feature from dxa, i.e. the adjoint of `.tile(1, HW, 1)` (training.py:171).
 * work: S*32*c floats.  Fixed summation order. */
int s2l_segment_colsums(const float* src, int ld, int c, int64_t rows_per_segment, int64_t n_segments, float* work,
                        float* out, s2l_stream_t stream);
/* Gradients of the tensors behind the pack-time fold of the first / skip layer (G = Wf [Wuv|Wa|Wt], c = Wf (buv+ba+bt) + bf;
 * tf_nerf.py:252-258, :269-281 followed by pts_linears[0] / the left half of pts_linears[5]): from dG [256,128] and dc [256]
 * (the weight / bias gradients s2l_wgrad returns for the folded layer) to d_first (the gradient of Wf, written to columns 0..255
 * of a [256, ld_first] tensor; `right` [256,256], when given with ld_first 512, is copied to columns 256..511), d_w_uv [256,42],
 * d_w_audio [256,64], d_w_time [256,20] and d_bias [256] (the common gradient of the three fc biases). */
int s2l_unfold_first_layer(const float* dG, const float* dc, const float* first_w, int ld_first, const float* w_uv,
                           const float* w_audio, const float* w_time, const float* b_uv, const float* b_audio,
                           const float* b_time, const float* right, float* d_first, float* d_w_uv, float* d_w_audio,
                           float* d_w_time, float* d_bias, s2l_stream_t stream);
/* Audio-encoder backward (autograd of tf_nerf.py:197-213): windows [B,16,29], dfeat [B,64] ->
 * grads [s2l_audio_grad_floats()]: encoder_conv.{0,2,4,6}.{weight,bias}, encoder_fc1.{0,2}.{weight,bias}
 * concatenated in that order, torch layouts.  work: ceil(B/4) * s2l_audio_grad_floats() floats. */
int64_t s2l_audio_grad_floats(void);
int s2l_audio_backward(const float* packed, const float* windows, const float* dfeat, float* work,
                       float* grads, int64_t n_windows, s2l_stream_t stream);
/* loss = weight * mean((pred - target)^2) over n_elems (training.py:605-619); dpred (optional) =
 * d loss / d pred; work: 1024 floats; loss: 1 float (device). */
int s2l_mse(const float* pred, const float* target, float weight, float* dpred, float* work,
            float* loss, int64_t n_elems, s2l_stream_t stream);

/* optimizer.step() of the reference's training loop (train.py:173-199 builds torch.optim.Adam; training.py:559-574 steps it after
 * check_weights) for EVERY tensor of a parameter group in ONE launch: torch's _single_tensor_adam arithmetic, operation for operation in
 * fp32 (no amsgrad, no maximize).  table: n_tensors records of four device pointers {param, grad, exp_avg, exp_avg_sq} (32 bytes each, fp32
 * contiguous tensors); counts[t]: elements of tensor t; blocks: n_blocks records {int32 tensor, int32 first element}, one per workgroup,
 * covering tensor t in pieces of s2l_adam_chunk() elements; the hyper-parameters are doubles, as torch holds them; step >= 1: the step number AFTER the increment (bias corrections 1 - beta^step).
 * nan_flags: NULL or n_tensors ints, set to 1 (never cleared) when the parameter as read, i.e. BEFORE this update, holds a NaN -- the
 * reference's check_weights (src/common.py:56-64, called at training.py:572) folded into the pass. */
int64_t s2l_adam_chunk(void);
int s2l_adam_step(const void* table, const void* blocks, const int64_t* counts, int64_t n_tensors, int64_t n_blocks, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, int* nan_flags, s2l_stream_t stream);

/* Paste + head-pose warp composite, up to but not including the U-Net
 * (TalkingFace.post_fusion2_onlylip_light, tf_nerf.py:320-386):
 *   merged_c = mask * pad(lip) + (1-mask) * face_canon                        (:339-352)
 *   M        = expanded rectangle rows [y0-p, y0+h+2p) x cols [x0-p, x0+w+p) (:354-364), or `mask`
 *              itself when expand_pad < 0
 *   out      = (gs(M) != 0) ? gs(merged_c) : rgb_gt     gs = bilinear, zeros, align_corners=False
 * lip [F,h,w,3]; face_canon, mask [FH,FW,3] when *_stride == 0 (per-clip constants) or
 * [F,FH,FW,3] when stride == FH*FW*3; rgb_gt [F,FH,FW,3]; coord [F,FH,FW,2];
 * out_new [F,FH,FW,3]; out_canonical [F,FH,FW,3] or NULL (rgb_merged_canonical, :352).
 * bgm: NULL, or the per-clip table of s2l_composite_tables for these face_canon/mask (used only
 * when both strides are 0): halves the gather traffic, results are bit-identical.
 * expand_pad = p (lip_w/5, or lip_w/12 for obama2, :357-360); pad_mode S2L_PAD_*.
 * Edge geometry follows the reference statement for statement (golden G17): a lip box that leaves the face frame is pasted
 * CROPPED (F.pad with negative amounts, :343-350); the rectangle is a python slice (:362) -- a negative bound wraps once, both
 * bounds are clamped, start >= stop is empty (then out = rgb_gt everywhere); a box that would need a crop larger than the lip
 * (beyond the frame by more than touching it) makes F.pad raise there and returns S2L_E_GEOMETRY here. */
int s2l_composite(const float* lip, const float* face_canon, int64_t face_stride, const float* mask,
                  int64_t mask_stride, const float* rgb_gt, const float* coord, float* out_new,
                  float* out_canonical, const float* bgm, int lip_h, int lip_w, int face_h, int face_w,
                  int x0, int y0, int pad_mode, int expand_pad, int64_t n_frames, s2l_stream_t stream);

/* Training branch of the same composite (post_fusion2_onlylip_light with use_post_fusion_blackaug=True, tf_nerf.py:371-384;
 * called so at training.py:436/445): hole1, hole2 [F,FH,FW] are the two N(0,1) fields `add_black_hole` (:306-318) draws with
 * torch.randn (channel 0 of a randn of the image shape), passed in by the caller WHEN the coin `random.random() > 0.5` of :371
 * came up; both NULL = the inference branch (s2l_composite is exactly that).  A hole is punched where the draw is < 1e-6
 * inside mask_face_observed = (grid_sample(face_canon > 0, coord) == 1): hole pixels of the merged image show rgb_gt and hole
 * pixels of rgb_gt show the merged image. */
int s2l_composite_train(const float* lip, const float* face_canon, int64_t face_stride, const float* mask,
                        int64_t mask_stride, const float* rgb_gt, const float* coord, const float* hole1,
                        const float* hole2, float* out_new, float* out_canonical, const float* bgm, int lip_h, int lip_w,
                        int face_h, int face_w, int x0, int y0, int pad_mode, int expand_pad, int64_t n_frames,
                        s2l_stream_t stream);
/* d lip of either branch: what loss.backward() (training.py:559) propagates from rgb_merged_new into rgb_lip_warped through
 * tf_nerf.py:339-386 (F.pad, the mask lerp, F.grid_sample, the blends).  d_new [F,FH,FW,3] -> d_lip [F,h,w,3] (overwritten).
 * The scatter into the lip box uses hardware float atomics: the summation order is not fixed (as ATen's grid_sample backward). */
int s2l_composite_backward_lip(const float* d_new, const float* face_canon, int64_t face_stride, const float* mask,
                               int64_t mask_stride, const float* coord, const float* hole1, const float* hole2,
                               float* d_lip, int lip_h, int lip_w, int face_h, int face_w, int x0, int y0, int pad_mode,
                               int expand_pad, int64_t n_frames, s2l_stream_t stream);

/* The inference composite of a CLIP at streaming speed: same result as s2l_composite (bit-identical out_new) for the common
 * case -- per-clip face_canon / mask folded into `bgm` (s2l_composite_tables), expanded-rectangle mask (expand_pad >= 0), no
 * black holes, no canonical output, face_h*face_w a multiple of 4, 16-byte-aligned frame streams.  Two kernels over spans of 256
 * pixels: quarter spans none of whose bilinear taps can reach the rectangle are copied rgb_gt -> out as 16-byte vectors (about
 * 77 % of a 500x500 frame with a 128x128 lip); the rest are evaluated one pixel per lane, every tap one aligned 16-byte gather
 * (the merged lip box of each frame is materialised first, 16 B per lip pixel).  n_frames <= 65535.
 * work: s2l_composite_stream_work_bytes(...) bytes, 16-byte aligned (span flags + merged lip boxes). */
int64_t s2l_composite_stream_work_bytes(int lip_h, int lip_w, int face_h, int face_w, int64_t n_frames);
int s2l_composite_stream(const float* lip, const float* mask, const float* bgm, const float* rgb_gt, const float* coord,
                         float* out_new, void* work, int lip_h, int lip_w, int face_h, int face_w, int x0, int y0,
                         int pad_mode, int expand_pad, int64_t n_frames, s2l_stream_t stream);

/* Optional per-clip precompute for s2l_composite: bgm [FH,FW,4] = ((1-mask)*face_canon, bits of face_canon > 0), the
 * background term of tf_nerf.py:352 as one 16-byte-aligned gather target (16-byte aligned). */
int s2l_composite_tables(const float* face_canon, const float* mask, float* bgm, int face_h, int face_w,
                         s2l_stream_t stream);

/* ---- post-fusion U-Net: eval-mode SimpleUnetLight --------------------------------
 * Replaces src/face_simple/models/SimpleUnetLight.py:99-111 as called at tf_nerf.py:387.
 * s2l_unet_pack: tensors_host = HOST array of 52 DEVICE pointers: for each of the ten 3x3 convolutions
 * in execution order (inc.0, inc.3, down1.0, down1.3, down2.0, down2.3, up1.0, up1.3, up2.0, up2.3)
 * {conv.weight [co,ci,3,3], bn.weight, bn.bias, bn.running_mean, bn.running_var}, then
 * outc.conv.weight [3,64,1,1], outc.conv.bias [3].  BatchNorm (eps = bn_eps) is folded into the packed
 * weights.  s2l_unet_forward: x [F,H,W,3] NHWC -> out [F,H,W,3]; H, W >= 4;
 * work: s2l_unet_work_floats(H, W, F) floats of scratch for the activations. */
int64_t s2l_unet_packed_floats(void);
int64_t s2l_unet_work_floats(int height, int width, int64_t n_frames);
int s2l_unet_pack(const float* const* tensors_host, float bn_eps, float* packed, s2l_stream_t stream);
int s2l_unet_forward(const float* packed, const uint16_t* packed16, const float* x, float* work, float* out, int height,
                     int width, int64_t n_frames, s2l_stream_t stream);   /* packed16: NULL = exact fp32 (see s2l_unet_pack16 below) */
/* The fp32 3x3 convolutions exist twice: as generated gfx950 assembly (csrc/gen_conv_body.py; the default wherever a launch has
 * the shape it takes) and as the C++ kernel it replaced.  Both perform the same arithmetic in the same order: outputs, input
 * gradients and weight gradients are bit-identical, which tests/test_gpu_unet_kernels.py checks over frame shapes by switching
 * here.  kind: 0 = assembly where available, 1 = the C++ kernel everywhere.  Process-wide (an atomic), a validation aid. */
int s2l_set_unet_conv_kernel(int kind);

/* Training: the same network keeping every activation (saved: s2l_unet_saved_floats(H, W, F) floats), and
 * its INPUT gradient d_out [F,H,W,3] -> d_x [F,H,W,3] (work: s2l_unet_backward_work_floats(H, W, F) floats of scratch) -- what
 * autograd propagates through the frozen eval-mode post_fusion_unet once `it > 100000` (train.py:188-197) from the sync loss
 * (training.py:491-557) and the face photometric loss (:458-459) back to the composite.  Needs the pack of this library
 * version (s2l_unet_pack also writes the transposed, tap-mirrored chunks the input-gradient convolutions read). */
int64_t s2l_unet_saved_floats(int height, int width, int64_t n_frames);
int64_t s2l_unet_backward_work_floats(int height, int width, int64_t n_frames);
int s2l_unet_forward_saved(const float* packed, const float* x, float* saved, float* out, int height, int width,
                           int64_t n_frames, s2l_stream_t stream);
int s2l_unet_backward(const float* packed, const float* saved, const float* d_out, float* work, float* d_x, int height,
                      int width, int64_t n_frames, s2l_stream_t stream);
/* The same pair on a WINDOW of the frame: x / out / d_out / d_x are [F,height,width,3] crops whose top-left corner sits at
 * (origin_y, origin_x) of a full_h x full_w frame (origins multiples of 4; sizes multiples of 4 unless the crop ends at the frame
 * edge).  The sync loss only reads the canonical-face box of the U-Net output (training.py:541-544), so the chain runs the network
 * on that box dilated by the network's dependency radius instead of the whole frame.  The crop is processed like a frame, except
 * that the align_corners=True up-samplings take their source positions from the FULL frame's geometry: every output and gradient
 * whose dependency cone (radius <= 32 pixels) stays inside the crop equals the full-frame value bit for bit; values nearer than
 * that to a crop edge that is not a frame edge must not be used. */
int s2l_unet_forward_saved_window(const float* packed, const uint16_t* packed16, const float* x, float* saved, float* out,
                                  int height, int width, int full_h, int full_w, int origin_y, int origin_x, int64_t n_frames,
                                  s2l_stream_t stream);
int s2l_unet_backward_window(const float* packed, const uint16_t* packed16, const float* saved, const float* d_out, float* work,
                             float* d_x, int height, int width, int full_h, int full_w, int origin_y, int origin_x,
                             int64_t n_frames, s2l_stream_t stream);
/* packed16 (or NULL): the nine 3x3 layers in bf16 operand form, s2l_unet_packed16_halves() uint16 written by s2l_unet_pack16
 * (same tensor table and BatchNorm fold as s2l_unet_pack; bn_eps < 0: NO fold, the raw weights -- the bf16 twin of s2l_unet_pack_raw,
 * for s2l_unet_train_*_bf16).  With it those convolutions and their input-gradient twins run on
 * v_mfma_f32_32x32x16_bf16 -- bf16 weights and staged inputs, fp32 accumulation, fp32 tensors in HBM -- the precision BASELINE
 * config 5 names for the training step; NULL = exact fp32 everywhere.  (The 3->64 first layer, 0.5 % of the work, stays fp32.) */
int64_t s2l_unet_packed16_halves(void);
int s2l_unet_pack16(const float* const* tensors_host, float bn_eps, uint16_t* packed16, s2l_stream_t stream);
/* Split operand form of the nine 3x3 layers for INFERENCE (SimpleUnetLight.py:16-111 as called at tf_nerf.py:387): every fp32 operand is
 * carried as two 16-bit parts, hi and lo = x - hi, and a product is evaluated as a_lo b_hi + a_hi b_lo + a_hi b_hi on a 16-bit MFMA
 * (three MFMAs at 16x the fp32-MFMA rate) with fp32 accumulation.  Since round 4 the parts are IEEE halves (hi = f16(x) toward zero,
 * lo = f16(x - hi) to nearest; v_mfma_f32_32x32x16_f16): 11 + 11 significant bits, 112 dB / RMSE 2.5e-6 against the exact fp32 network
 * (bf16 parts, rounds 2-3: 99 dB; plain bf16: 46 dB); operands beyond +-65504 saturate.  The result stays inside the north-star
 * tolerance (PSNR >= 50 dB / RMSE <= 1e-4) by orders of magnitude; the exact fp32 kernels remain the default.  s2l_unet_pack16x3 writes
 * s2l_unet_packed16x3_halves() uint16 (same tensor table and BatchNorm fold as s2l_unet_pack); s2l_unet_forward_split = s2l_unet_forward
 * with those layers in this form. */
int64_t s2l_unet_packed16x3_halves(void);
int s2l_unet_pack16x3(const float* const* tensors_host, float bn_eps, uint16_t* packed16x3, s2l_stream_t stream);
int s2l_unet_forward_split(const float* packed, const uint16_t* packed16x3, const float* x, float* work, float* out, int height,
                           int width, int64_t n_frames, s2l_stream_t stream);
/* Which kernel runs the split-bf16 layers: 0 (default) the persistent form with a two-chunk-deep operand pipeline, 1 the
 * one-tile-per-workgroup form (the persistent kernel's fall-back for launches it does not take), 2 the generated-assembly form
 * (csrc/conv16.hip; layers it does not cover run as 0) -- form 2 exists only in libs2l_hip_ref.so, the test-side build with
 * -DS2L_WITH_REFERENCE_KERNELS; the product library answers S2L_E_UNSUPPORTED.
 * Same arithmetic in the same order: the outputs are the same bits (a test aid).  Any other value: S2L_E_SIZE. */
int s2l_set_unet_split_kernel(int kind);

/* TRAIN mode of the same network, as the reference runs it until `it > 100000` (train.py:188-197): every BatchNorm2d normalises
 * with the statistics of the batch (biased variance) and updates its running statistics in place (momentum, unbiased variance:
 * nn.BatchNorm2d), the weights receive gradients (loss.backward(), training.py:559, through tf_nerf.py:387).
 * s2l_unet_pack_raw: the RAW (un-folded) weights in the kernels' chunk layout, forward and transposed; same table and blob size as
 *   s2l_unet_pack; re-run after every optimizer step.
 * s2l_unet_train_forward: tensors_host = the s2l_unet_pack table (its running_mean / running_var entries are WRITTEN when
 *   update_running != 0); saved: s2l_unet_train_saved_floats(H, W, F) floats; scratch: 262144 floats.
 * s2l_unet_train_backward: d_out [F,H,W,3] -> d_x [F,H,W,3] (or NULL) and grads [s2l_unet_grad_floats()]: for each of the ten 3x3
 *   layers in execution order conv.weight [cout,cin,3,3], bn.weight [cout], bn.bias [cout]; then outc.conv.weight [3,64],
 *   outc.conv.bias [3].  work: s2l_unet_train_work_floats(H, W, F) floats.  Weight gradients are split-K MFMA GEMMs over the
 *   pixels, reduced in a fixed order.  grads == NULL (with d_x != NULL): a FROZEN net that still runs train-mode BatchNorm (the
 *   reference after it > 100000, where Trainer.train_step's model.train() undoes train.py:195's .eval()): only d_x, the weight-
 *   gradient kernels are not launched. */
int64_t s2l_unet_train_saved_floats(int height, int width, int64_t n_frames);
int64_t s2l_unet_train_work_floats(int height, int width, int64_t n_frames);
int64_t s2l_unet_grad_floats(void);
int s2l_unet_pack_raw(const float* const* tensors_host, float* packed, s2l_stream_t stream);
int s2l_unet_train_forward(const float* packed_raw, const float* const* tensors_host, float bn_eps, float momentum,
                           int update_running, const float* x, float* saved, float* scratch, float* out, int height, int width,
                           int64_t n_frames, s2l_stream_t stream);
int s2l_unet_train_backward(const float* packed_raw, const float* const* tensors_host, const float* x, const float* saved,
                            const float* d_out, float* work, float* d_x, float* grads, int height, int width, int64_t n_frames,
                            s2l_stream_t stream);
/* The same two passes with the 3x3 layers 1..9 (forward convolutions; their input-gradient twins) on bf16 operands, for the bf16
 * training step of BASELINE config 5 when the frozen net runs train-mode BatchNorm (the reference's loop, training.py:150):
 * packed16_raw = s2l_unet_pack16 called with bn_eps < 0 (no fold: the raw weights, forward and transposed halves).  Accumulation,
 * tensors, batch statistics, the BatchNorm backward, the weight gradients and the 3->64 first layer stay fp32. */
int s2l_unet_train_forward_bf16(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host, float bn_eps,
                                float momentum, int update_running, const float* x, float* saved, float* scratch, float* out,
                                int height, int width, int64_t n_frames, s2l_stream_t stream);
int s2l_unet_train_backward_bf16(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                 const float* x, const float* saved, const float* d_out, float* work, float* d_x, float* grads,
                                 int height, int width, int64_t n_frames, s2l_stream_t stream);
/* F successive ONE-FRAME train-mode calls in one set of launches -- how the reference's loop runs the (frozen) net: tf_nerf.py:387
 * inside train_stage1's batch-1 calls, main frame then the five window frames of every sample (training.py:436-459, 504-548), with
 * Trainer.train_step's model.train() in force (training.py:150).  Every frame is its own statistics group: normalised with its own
 * batch statistics; the running statistics move once per frame, in frame order; the BatchNorm backward carries each frame's own
 * terms.  Per frame the arithmetic is that of a call with n_frames = 1: the same bits (tests: test_unet_train_frames_*).
 * packed16_raw: NULL = fp32 convolutions, else the s2l_unet_pack16(bn_eps < 0) blob.  The backward returns the input gradient
 * only (a frozen net).  Sizes: s2l_unet_train_frames_{saved,scratch,work}_floats. */
int64_t s2l_unet_train_frames_saved_floats(int height, int width, int64_t n_frames);
int64_t s2l_unet_train_frames_scratch_floats(int64_t n_frames);
int64_t s2l_unet_train_frames_work_floats(int height, int width, int64_t n_frames);
int s2l_unet_train_forward_frames(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host, float bn_eps,
                                  float momentum, int update_running, const float* x, float* saved, float* scratch, float* out,
                                  int height, int width, int64_t n_frames, s2l_stream_t stream);
int s2l_unet_train_backward_frames(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                   const float* x, const float* saved, const float* d_out, float* work, float* d_x, int height,
                                   int width, int64_t n_frames, s2l_stream_t stream);

/* s2l_unet_train_backward_frames for a net that still TRAINS (before `it > 100000`, train.py:188-197 not reached): also the parameter
 * gradients of the F one-frame calls, summed over the frames (layout and meaning of `grads` as s2l_unet_train_backward; d_x may be NULL). */
int s2l_unet_train_backward_frames_grads(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                         const float* x, const float* saved, const float* d_out, float* work, float* d_x, float* grads,
                                         int height, int width, int64_t n_frames, s2l_stream_t stream);

/* The same pair on HALF-WIDTH TENSORS (csrc/unet_half.inc, csrc/convh.hip): every tensor between the kernels -- pre-BatchNorm outputs,
 * activations, pooled / up-sampled copies, gradients -- is bf16 NHWC; bf16 operands, fp32 accumulation, fp32 per-frame statistics (of
 * the bf16-rounded pre-BatchNorm tensor), results rounded to nearest even on store; x, out, d_out, d_x stay fp32.  Replaces the same
 * reference lines as s2l_unet_train_forward_frames / _backward_frames (the frozen net of training.py:436-459 left in .train() by
 * Trainer.train_step: tf_nerf.py:387 -> SimpleUnetLight.py:99-111 with batch statistics per one-frame call) at about half their HBM and
 * texture-path traffic.  packed16_raw: s2l_unet_pack16 with bn_eps < 0 (required).  saved: ..._h_saved_halves halves; scratch:
 * ..._h_scratch_floats floats; work: ..._h_work_halves halves.  A geometry the convolution kernel does not take: S2L_E_SIZE. */
int64_t s2l_unet_train_frames_h_saved_halves(int height, int width, int64_t n_frames);
int64_t s2l_unet_train_frames_h_scratch_floats(int64_t n_frames);
int64_t s2l_unet_train_frames_h_work_halves(int height, int width, int64_t n_frames);
int s2l_unet_train_forward_frames_h(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host, float bn_eps,
                                    float momentum, int update_running, const float* x, uint16_t* saved, float* scratch, float* out,
                                    int height, int width, int64_t n_frames, s2l_stream_t stream);
/* ... for a FROZEN net (the backward is s2l_unet_train_backward_frames_h: no weight gradients): the BatchNorm + ReLU passes whose only
 * reader is the next 3x3 convolution at the same resolution (a0, a2, a4, a6, a8 of SimpleUnetLight.py:16-111's DoubleConvs) are folded into
 * that convolution -- it reads the pre-BatchNorm tensor and normalises its halo tiles in LDS with the per-frame scale / shift, the same
 * expression and rounding --, a5 and a7 are formed inside the up-sampling that reads them, so `out`, every stored z and the statistics are
 * the bits of s2l_unet_train_forward_frames_h; those seven activations' slots of `saved` are left unwritten (the frozen net's backward
 * re-forms the two masks it needs from z). */
int s2l_unet_train_forward_frames_h_fused(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host, float bn_eps,
                                          float momentum, int update_running, const float* x, uint16_t* saved, float* scratch, float* out,
                                          int height, int width, int64_t n_frames, s2l_stream_t stream);
int s2l_unet_train_backward_frames_h(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                     const uint16_t* saved, const float* d_out, uint16_t* work, float* d_x, int height, int width,
                                     int64_t n_frames, s2l_stream_t stream);
/* s2l_unet_train_backward_frames_h for a net that still TRAINS (before `it > 100000`): also the parameter gradients of the F one-frame
 * calls, summed over the frames (layout of `grads` as s2l_unet_train_backward); the 3x3 layers' weight gradients take bf16 operands
 * straight from the bf16 planes.  x: the forward's input [F,H,W,3]; d_x may be NULL.  work: s2l_unet_train_frames_h_work_halves_grads. */
int64_t s2l_unet_train_frames_h_work_halves_grads(int height, int width, int64_t n_frames);
int s2l_unet_train_backward_frames_h_grads(const float* packed_raw, const uint16_t* packed16_raw, const float* const* tensors_host,
                                           const float* x, const uint16_t* saved, const float* d_out, uint16_t* work, float* d_x, float* grads,
                                           int height, int width, int64_t n_frames, s2l_stream_t stream);
/* The EVAL-mode pair (s2l_unet_forward_saved_window / s2l_unet_backward_window with bf16 operands) on half-width tensors: BatchNorm
 * folded into the weights (packed16: s2l_unet_pack16 with the real eps; packed: s2l_unet_pack's blob for the first layer, the biases and
 * the output layer), bias + ReLU in the convolution's epilogue, every tensor between the kernels bf16 in 32-channel planes; window
 * semantics and geometry errors as the fp32-tensor pair.  saved: s2l_unet_saved_h_halves halves; work: s2l_unet_backward_h_work_halves. */
int64_t s2l_unet_saved_h_halves(int height, int width, int64_t n_frames);
int64_t s2l_unet_backward_h_work_halves(int height, int width, int64_t n_frames);
int s2l_unet_forward_saved_h(const float* packed, const uint16_t* packed16, const float* x, uint16_t* saved, float* out, int height, int width,
                             int full_h, int full_w, int origin_y, int origin_x, int64_t n_frames, s2l_stream_t stream);
int s2l_unet_backward_h(const float* packed, const uint16_t* packed16, const uint16_t* saved, const float* d_out, uint16_t* work, float* d_x,
                        int height, int width, int full_h, int full_w, int origin_y, int origin_x, int64_t n_frames, s2l_stream_t stream);
/* One 3x3 layer (1..9) of that chain on its own: out [F,H,W,cout] bf16 = conv(concat(inA, inB) bf16, the layer's raw bf16 weights);
 * transposed != 0: the layer's input gradient (inA = the gradient of its output; gate, or NULL: [F,H,W,cin] bf16, out = 0 where
 * gate <= 0).  s2l_debug_conv_layer_f32: the fp32-tensor kernel it replaces (same operands, same accumulation order: on
 * bf16-representable inputs s2l_convh_layer's output is the round-to-nearest-even bf16 of its output -- the tests' comparator). */
int s2l_convh_layer(const uint16_t* packed16_raw, int layer, int transposed, const uint16_t* inA, int CA, const uint16_t* inB, int CB,
                    const uint16_t* gate, uint16_t* out, int height, int width, int64_t n_frames, s2l_stream_t stream);
int s2l_debug_conv_layer_f32(const float* packed_raw, const uint16_t* packed16_raw, int layer, int transposed, const float* inA, int CA,
                             const float* inB, int CB, const float* gate, float* out, int height, int width, int64_t n_frames,
                             s2l_stream_t stream);

/* The 3x3 weight gradient of the half-width training chain on its own (a test aid; the reference's is autograd's conv2d weight gradient
 * inside SimpleUnetLight.py:16-111's layers): dz [F][cout/32][H][W][32] bf16 planes, inputs inA (CA channels) and optionally inB (CB) as
 * planes -- CA, CB, cout multiples of 64 --, dw [cout][CA + CB][9] fp32 (tap = 3 ky + kx), partial: 64*256*128*9 floats of scratch. */
int s2l_debug_conv_wgrad_h(const uint16_t* dz, const uint16_t* inA, int CA, const uint16_t* inB, int CB, int cout, float* partial, float* dw,
                           int height, int width, int64_t n_frames, s2l_stream_t stream);

/* Test aid: a forward 3x3 layer of the half-width chain that also leaves its tiles' partial sums of BatchNorm's batch statistics of the
 * STORED (bf16-rounded) values -- stat[((frame * blocks + block) * 2 + {sum, sum of squares}) * cout + channel], *blocks_out blocks per frame
 * (0: the kernel form that ran does not leave them) -- as s2l_unet_train_forward_frames_h consumes them in place of a pass over the tensor
 * (nn.BatchNorm2d in train mode: SimpleUnetLight.py:16-40).  stat: n_frames * 1024 * 2 * cout floats. */
int s2l_debug_convh_layer_stats(const uint16_t* packed16_raw, int layer, const uint16_t* inA, int CA, const uint16_t* inB, int CB,
                                uint16_t* out, float* stat, int* blocks_out, int height, int width, int64_t n_frames, s2l_stream_t stream);

/* Test aid: the INPUT-gradient convolution of a layer of the half-width chain (dz: cout channels in, cin channels out) that also leaves stage 1
 * of the BatchNorm backward of the layer below it -- z: that layer's pre-BatchNorm tensor, st_rows: its per-frame rows of 512 floats (scale at
 * [c], shift at [cin + c]) --: stat[((frame * blocks + block) * 2 + k) * cin + channel] with k = 0: sum g', k = 1: sum g' z, where
 * g' = fma(z, scale, shift) > 0 ? g : 0 on the stored (bf16) g; out itself is stored unmasked.  What s2l_unet_train_backward_frames_h runs in
 * place of its reduction pass over g and z (autograd's BatchNorm backward: SimpleUnetLight.py:16-40).  stat: n_frames * 1024 * 2 * cin floats. */
int s2l_debug_convh_layer_bstats(const uint16_t* packed16_raw, int layer, const uint16_t* dz, const uint16_t* z, const float* st_rows,
                                 uint16_t* out, float* stat, int* blocks_out, int height, int width, int64_t n_frames, s2l_stream_t stream);

/* Measurement aid (tools/ubench_mfma.py): `waves` (4 or 8) waves per CU each issue iters x 8 independent v_mfma_f32_32x32x16_bf16 on
 * registers and nothing else -- the rate the chip sustains under that load (the clock drops below its 2.4 GHz peak). */
int s2l_debug_bf16_mfma_rate(int64_t iters, int waves, float* sink, s2l_stream_t stream);
/* Which form of the half-width convolution runs: 0 (default) eight waves per workgroup (two per SIMD), each interleaving its loads with its
 * MFMAs, 1 four waves (one per SIMD), 2 eight waves with the two waves of a SIMD alternating between an MFMA-only segment and a load /
 * epilogue segment (csrc/gen_convhx_body.py; launches with a gate input run as 0) -- measured equal to form 0 within +-3 %: the kernel is
 * bound by the CU's vector-memory path, not by its instruction schedule (docs/LABNOTES.md §10).  Forms 1 and 2 exist only in
 * libs2l_hip_ref.so (-DS2L_WITH_REFERENCE_KERNELS: tests and tools/soak_conv_kernels.py load it); the product library answers
 * S2L_E_UNSUPPORTED.  Same arithmetic in the same order: the outputs are the same bits.  Any other value: S2L_E_SIZE. */
int s2l_set_unet_half_kernel(int kind);

/* Crop + bilinear resize between the U-Net and the sync expert, and its adjoint (training.py:541-544:
 * rgb_merged[:, y:y2, x:x2, :] then transforms.Resize([96,96]); torchvision 0.9.0 resizes tensors with
 * F.interpolate(mode='bilinear', align_corners=False), no antialiasing).  src [F,src_h,src_w,3]; box = data['canonical_face_bbox'];
 * dst [F,out_h,out_w,3] when window_t == 0, or the rgb_window layout [F/T,3,T,out_h,out_w] (:547-548) with frame f = s*T + t
 * when window_t == T > 0.  s2l_crop_resize_backward: d_dst (same layout) -> d_src [F,src_h,src_w,3], zero outside the box
 * (deterministic gather, no atomics).  x2 / y2 beyond the frame are clipped to it, as the python slice clips them (the scale is
 * that of the clipped crop); x < 0, y < 0 or an empty box return S2L_E_GEOMETRY. */
int s2l_crop_resize(const float* src, int src_h, int src_w, int x, int y, int x2, int y2, float* dst, int out_h, int out_w,
                    int window_t, int64_t n_frames, s2l_stream_t stream);
int s2l_crop_resize_backward(const float* d_dst, int src_h, int src_w, int x, int y, int x2, int y2, float* d_src, int out_h,
                             int out_w, int window_t, int64_t n_frames, s2l_stream_t stream);

/* ---- pose -> warp grid -----------------------------------------------------------
 * s2l_rel_pose replaces prepare_transform_matrix + compute_rel_pose* (src/face_simple/models/utils.py:36-77;
 * Trainer.compute_rel_pose*, src/face_simple/training.py:263-275): euler, trans [F,3]; canon_euler,
 * canon_trans [3] (data['canonical_euler'/'canonical_trans']); T [F,16] row-major 4x4; mode S2L_POSE_*.
 * s2l_warp_grid replaces BackprojectDepth.forward + Project3D.forward (utils.py:131-169) with
 * K = [[focal,0,W/2],[0,focal,H/2],[0,0,1]] (training.py:298-302): depth [H,W] when depth_stride == 0 (one
 * canonical depth map for the clip) or [F,H,W] when depth_stride == H*W (per-frame depth, face_tracker.py:
 * 586-603); grid [F,H,W,2] in grid_sample units, clamped to [-1,1] when clamp != 0 (face_tracker.py:606);
 * z NULL or [F,H,W] = projected depth (Project3D return_z).
 * s2l_grid_sample: F.grid_sample(bilinear, align_corners=False) on NHWC 3-channel images, padding
 * S2L_SAMPLE_ZEROS or S2L_SAMPLE_BORDER (training.py:312): img [img_h,img_w,3] (img_stride 0) or
 * [F,img_h,img_w,3]; grid [F,out_h,out_w,2]; out [F,out_h,out_w,3]. */
int s2l_rel_pose(const float* euler, const float* trans, const float* canon_euler, const float* canon_trans,
                 int mode, float* T, int64_t n_frames, s2l_stream_t stream);
int s2l_warp_grid(const float* depth, int64_t depth_stride, const float* T, float focal, int clamp,
                  float* grid, float* z, int height, int width, int64_t n_frames, s2l_stream_t stream);
int s2l_grid_sample(const float* img, int64_t img_stride, const float* grid, float* out, int img_h, int img_w,
                    int out_h, int out_w, int padding, int64_t n_frames, s2l_stream_t stream);

/* Canonical-depth photometric loss (Trainer.inverse_warping + add_loss_canonical_depth_photo, src/face_simple/training.py:462-477,
 * 296-314, 621-634) and its gradient w.r.t. the depth map (model.canonical_depth_head, tf_nerf.py:174-195), which is what
 * loss.backward() leaves in canonical_depth_head.grad:
 *   pred = grid_sample(src, Project3D(BackprojectDepth(depth), K, T), padding_mode='border')      src = rgb_face_gt [F,H,W,3]
 *   loss = weights * sum((pred - target)^2 * mask) / (sum(mask) + 1e-6)     (mask NULL: weights * mean((pred - target)^2))
 * depth [H,W]; T [F,16] (compute_rel_pose_inverse); target (rgb_face_canonical) and mask [H,W,3] when *_stride == 0, else
 * [F,H,W,3]; loss: TWO floats (the loss, then an internal scale); d_depth [H,W] or NULL; work:
 * s2l_depth_photo_work_floats(H, W) floats.  Each output pixel depends on depth at that pixel only: no scatter. */
int64_t s2l_depth_photo_work_floats(int height, int width);
int s2l_depth_photo_loss(const float* depth, const float* T, float focal, const float* src, const float* target,
                         int64_t target_stride, const float* mask, int64_t mask_stride, float weights, float* work,
                         float* loss, float* d_depth, int height, int width, int64_t n_frames, s2l_stream_t stream);

/* ---- T3: lip-sync expert loss ---------------------------------------------------
 * Replaces SyncNet_color.forward (src/face_simple/models/syncnet.py:57-67, conv.py:5-19) in eval mode, and
 * Trainer.cosine_loss / get_sync_contrastive_loss (src/face_simple/training.py:576-603) with the gradient the
 * reference gets from autograd for the generated window (the net is frozen, training.py:85-90).
 * s2l_syncnet_pack: tensors_host = HOST array of 31*6 DEVICE pointers: for the 17 face_encoder blocks then the
 * 14 audio_encoder blocks {conv_block.0.weight [co,ci,kh,kw], conv_block.0.bias, conv_block.1.weight, .bias,
 * .running_mean, .running_var}; BatchNorm (eps = bn_eps) is folded.
 * s2l_syncnet_forward: mel [B,80,16] (= [B,1,80,16]); face [B,48,96,15] NHWC (channel 3t+c, BGR, lower half
 * rows: build it with s2l_sync_window); audio_emb, face_emb [B,512], L2-normalised (:63-64); work:
 * s2l_syncnet_work_floats(B) floats, keeps the activations s2l_syncnet_face_backward needs.
 * s2l_sync_loss: BCELoss(cosine_similarity(a, v), y) * weight (mean over B); y [B]; scratch [B]; *loss is
 * overwritten, or added to when accumulate != 0; d_face_emb NULL or [B,512] = d loss / d face_emb.
 * s2l_syncnet_face_backward: d_face [B,48,96,15] = d loss / d face, from d_face_emb and the `work` of the forward.
 * s2l_sync_window: g_rgb [B,3,T,H,W] (RGB, the reference's rgb_window layout) -> face [B,H-H/2,W,3T]
 * (training.py:588-590); s2l_sync_window_backward is its adjoint (rows above H/2 get zero). */
int64_t s2l_syncnet_packed_floats(void);
int64_t s2l_syncnet_work_floats(int64_t batch);
int s2l_syncnet_pack(const float* const* tensors_host, float bn_eps, float* packed, s2l_stream_t stream);
int s2l_syncnet_forward(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                        float* face_emb, int64_t batch, s2l_stream_t stream);
int s2l_sync_loss(const float* audio_emb, const float* face_emb, const float* y, float weight, float* scratch,
                  float* loss, int accumulate, float* d_face_emb, int64_t batch, s2l_stream_t stream);
int s2l_syncnet_face_backward(const float* packed, const float* face, float* work, const float* d_face_emb,
                              float* d_face, int64_t batch, s2l_stream_t stream);
/* The contrastive loss embeds the generated AND the negative windows of the same audio (training.py:592-601).  s2l_syncnet_forward_pair:
 * face [face_batch,48,96,15] (generated windows first), mel [audio_batch,80,16], audio_batch <= face_batch: one pass per encoder --
 * twice the columns per weight read in the face encoder, the audio encoder once; work: s2l_syncnet_work_floats(face_batch).
 * s2l_syncnet_face_backward_prefix: d_face [batch,48,96,15] for the FIRST `batch` windows of a forward over work_batch windows. */
int s2l_syncnet_forward_pair(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                             float* face_emb, int64_t audio_batch, int64_t face_batch, s2l_stream_t stream);
int s2l_syncnet_face_backward_prefix(const float* packed, const float* face, float* work, const float* d_face_emb,
                                     float* d_face, int64_t batch, int64_t work_batch, s2l_stream_t stream);
/* Opt-in speed mode of the two entry points above (round 5; the exact fp32 convolutions stay the default and are what the parity
 * tests pin): the implicit-GEMM convolutions of 64-row tiles run with BOTH operands as hi + lo bf16 parts -- three
 * v_mfma_f32_16x16x32_bf16 per product block, fp32 accumulation, ~1e-5 relative on a layer's output; fp32 range, so the gradients
 * keep their small values (csrc/conv_gemm.h).  Same arguments, same `packed` blob (s2l_syncnet_pack writes the parts too), same
 * `work`; a backward must use the form its forward used. */
int s2l_syncnet_forward_pair_split(const float* packed, const float* mel, const float* face, float* work, float* audio_emb,
                                   float* face_emb, int64_t audio_batch, int64_t face_batch, s2l_stream_t stream);
int s2l_syncnet_face_backward_prefix_split(const float* packed, const float* face, float* work, const float* d_face_emb,
                                           float* d_face, int64_t batch, int64_t work_batch, s2l_stream_t stream);
int s2l_sync_window(const float* g_rgb, float* face, int n_frames_t, int height, int width, int64_t batch,
                    s2l_stream_t stream);
int s2l_sync_window_backward(const float* d_face, float* d_g_rgb, int n_frames_t, int height, int width,
                             int64_t batch, s2l_stream_t stream);

/* ---- perceptual loss: lpips.LPIPS(net='alex', version='0.1') as Trainer.add_perceptual_loss calls it
 * (src/face_simple/training.py:76, 655-674; requirement.txt:11 pins the third-party package lpips==0.1.4, whose published
 * forward pass csrc/lpips.hip restates) and the gradient the reference gets from autograd for the generated image.
 * s2l_lpips_pack: tensors_host = HOST array of 17 DEVICE pointers: conv1..conv5 {weight [co,ci,kh,kw], bias [co]}
 * (net.slice1.0, slice2.3, slice3.6, slice4.8, slice5.10), lin0..lin4 model.1.weight [1,C,1,1], scaling_layer.shift [3],
 * scaling_layer.scale [3].
 * s2l_lpips_forward: in0, in1 [N,H,W,3] NHWC in [-1,1], or in [0,1] with from01 != 0 (then (x - 0.5) * 2 is applied first,
 * as add_perceptual_loss does, training.py:669-670); out [N] = the package's [N,1,1,1] distance; work:
 * s2l_lpips_work_floats(H, W, N) floats, keeps the activations s2l_lpips_backward needs.
 * s2l_lpips_backward: d_out [N] -> d_in0 [N,H,W,3], overwritten or (accumulate != 0) added to; same from01 as the forward
 * (in1 is the ground truth: no gradient).
 * H, W >= 31 (AlexNet's two 3x3/2 poolings after the 11x11/4 convolution): smaller -> S2L_E_GEOMETRY. */
int64_t s2l_lpips_packed_floats(void);
int64_t s2l_lpips_work_floats(int height, int width, int64_t batch);
int s2l_lpips_pack(const float* const* tensors_host, float* packed, s2l_stream_t stream);
int s2l_lpips_forward(const float* packed, const float* in0, const float* in1, int from01, float* work, float* out,
                      int height, int width, int64_t batch, s2l_stream_t stream);
int s2l_lpips_backward(const float* packed, float* work, const float* d_out, int from01, int accumulate, float* d_in0,
                       int height, int width, int64_t batch, s2l_stream_t stream);
/* The same two calls with conv2..conv5 and their input gradients in the split-operand form (see s2l_syncnet_forward_pair_split;
 * conv1 and its input gradient keep their own exact kernels). */
int s2l_lpips_forward_split(const float* packed, const float* in0, const float* in1, int from01, float* work, float* out,
                            int height, int width, int64_t batch, s2l_stream_t stream);
int s2l_lpips_backward_split(const float* packed, float* work, const float* d_out, int from01, int accumulate, float* d_in0,
                             int height, int width, int64_t batch, s2l_stream_t stream);

/* ---- bf16 mode of the training step (BASELINE config 5; same mathematics as s2l_train_forward / _backward /
 * s2l_wgrad, operands and saved state in bf16, fp32 accumulation, fp32 master weights and gradients) -------------
 * Replaces, like the fp32 entry points, the autograd of Trainer.predict_lip_image + add_photometric_loss
 * (src/face_simple/training.py:158-251, 605-619) through TalkingFace.rgb_forward (tf_nerf.py:225-285).
 * s2l_pack_bf16: bf16 operand images from the state-dict tensors (table as s2l_pack_weights) and the fp32 blob of
 * s2l_pack_weights (folded first/skip matrices, biases); packed_bf16: s2l_bf16_packed_halves() uint16.
 * Rows are processed in tiles of 256: Np = s2l_bf16_rows_padded(N).  hT, dzT: bf16 [8][Np/32][8][2][64][8]
 * ([layer][group of 32 rows][32-feature block][half][lane n + 32 hh][4 (a & 1) + c] = feature 32R + 8a + 4hh + c of row n, half = a >> 1:
 * each of a block's two store instructions writes 1 KiB of contiguous memory, csrc/s2l_bf16.h); masks: Np * 32 bytes per layer ([8][Np/64][256] uint64 of storage) holding one ReLU mask dword per lane and stage (layout: csrc/s2l_bf16.h), written by the forward and read by the backward only; xT: the embedded
 * rows in the same image layout with 4 blocks, bf16 [Np/32][4][2][64][8] (s2l_ensemble_rows_bf16 for a whole batch of frames:
 * feat [F,64], time_index int64 [F] and u01 fp32 [F] on the device, areas fp32 [4*HW*F]; or s2l_rows_to_tiles_bf16 from
 * fp32 rows); rgb, drgb: fp32 [N,3]; dxa: fp32 [N,64]. */
int64_t s2l_bf16_packed_halves(void);
int64_t s2l_bf16_rows_padded(int64_t n_rows);
int s2l_pack_bf16(const float* const* tensors_host, const float* packed_f32, uint16_t* packed_bf16,
                  s2l_stream_t stream);
int s2l_ensemble_rows_bf16(const float* packed, const float* coords, const float* feat, const int64_t* time_index,
                           const float* u01, int width, int height, uint16_t* xT, float* areas, int64_t n_pixels,
                           int64_t n_frames, s2l_stream_t stream);
int s2l_train_forward_bf16(const uint16_t* packed_bf16, const float* packed_f32, const uint16_t* xT, uint16_t* hT,
                           uint64_t* masks, float* rgb, int64_t n_rows, s2l_stream_t stream);
/* The bf16 forward exists twice: a generated gfx950 assembly kernel (csrc/gen_fwd16_body.py: 64 rows per wave, one wave per
 * SIMD; the default) and the C++ kernel it replaced (32 rows per wave, two waves per SIMD), which performs the same arithmetic
 * in the same order -- images, masks and rgb are bit-identical (tests switch between the two).
 * kind: 0 = assembly (default), 1 = C++.  Process-wide (an atomic). */
int s2l_set_bf16_forward_kernel(int kind);
int s2l_train_backward_bf16(const uint16_t* packed_bf16, const float* drgb, const uint64_t* masks, uint16_t* dzT,
                            float* dxa, int64_t n_rows, s2l_stream_t stream);
/* The backward exists twice as well: the C++ kernel above and a generated gfx950 assembly kernel (csrc/gen_bwd16_body.py: 64 rows
 * per wave, one wave per SIMD) whose dz images are bit-identical.  The assembly kernel returns the gradient of the audio columns
 * summed over each 256-row tile -- dxa_tiles [ceil(n_rows / 256)][64], tile t = rows [256 t, 256 t + 256) -- instead of per row
 * (the per-row form costs 2.4 GB of traffic per 64-frame step only to be column-summed per frame, training.py:171's adjoint): use
 * it when a frame's 4 H W rows are a whole number of tiles, then s2l_segment_colsums over the tiles of each frame. */
int s2l_train_backward_bf16_tiles(const uint16_t* packed_bf16, const float* drgb, const uint64_t* masks, uint16_t* dzT,
                                  float* dxa_tiles, int64_t n_rows, s2l_stream_t stream);
/* Weight gradients from the tiles: dw [256,k_in] = dzT_layer^T . inT (k_in = 256: inT = the hT layer below; k_in = 128:
 * inT = the embedded rows in the same image layout, s2l_rows_to_tiles_bf16), db NULL or [256] = column sums of dz; work:
 * s2l_wgrad_bf16_work_floats() floats (per-workgroup partial sums, reduced in a fixed order).  s2l_out_grad_bf16:
 * dwout [3,256] = drgb^T h7, dbout [3] = column sums of drgb (output_linear). */
int64_t s2l_wgrad_bf16_work_floats(void);
int s2l_wgrad_bf16(const uint16_t* dzT_layer, const uint16_t* inT, int k_in, float* work, float* dw, float* db,
                   int64_t n_rows, s2l_stream_t stream);
int s2l_rows_to_tiles_bf16(const float* x, int k, uint16_t* xT, int64_t n_rows, s2l_stream_t stream);
int s2l_out_grad_bf16(const float* drgb, const uint16_t* h7T, float* work, float* dwout, float* dbout,
                      int64_t n_rows, s2l_stream_t stream);

/* ---- 8-bit output (inference.py:172-178) ---------------------------------------------------------------
 * out[i] = saturate_cast<uchar>(rgb[i] * 255) as cv2.imwrite converts the float image the reference hands it
 * (round to nearest even, clamp to [0,255]); n = number of floats.  Channel order is untouched (the reference's
 * RGB->BGR swap only undoes cv2's BGR file convention).  rgb needs 4-byte alignment only (a frame slice of a clip starts at
 * any multiple of H*W*3 floats), out none. */
int s2l_to8b(const float* rgb, uint8_t* out, int64_t n, s2l_stream_t stream);
/* The reader's conversion of decoded 8-bit images (someones_lip_dataset.py:196-217: array / 255. in float64, then float32):
 * out[i] = (float)((double)in[i] / 255.0), bit for bit what the host conversion yields, so that frames can cross PCIe as bytes.
 * in: 4-byte aligned, out: 16-byte aligned. */
int s2l_from8b(const uint8_t* in, float* out, int64_t n, s2l_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S2L_HIP_H */
