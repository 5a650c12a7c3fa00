/*
 * apa.h -- C ABI of libapa_hip.so, the MI355X (gfx950) attentional-pooling library.
 *
 * This is the drop-in boundary for the attentional-pooling hot path of
 * rohitgirdhar/AttentionalPoolingAction.  The reference has no native implementation of the
 * head (it is ~10 TF-slim graph ops, models/slim/nets/nets_factory.py:242-328); its only native
 * plugin mechanism is the set of TF custom ops in src/custom_ops/ (REGISTER_OP + OpKernel::Compute,
 * loaded by src/custom_ops/custom_ops_factory.py:11-18 via tf.load_op_library).  Each entry point
 * below names the reference code it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (APA_OK) or a negative apa_status.
 *     Nothing throws across the boundary (the reference signals errors with OP_REQUIRES_OK /
 *     assert, src/custom_ops/pose_to_heatmap.cc:27-32,49).
 *   - all tensor pointers are DEVICE pointers (HBM) unless the name ends in _host.
 *   - the caller owns every buffer, including workspaces; the library is stateless and
 *     re-entrant (the reference ops run concurrently on TF inter-op threads,
 *     src/custom_ops/pose_utils.hpp:7-14).
 *   - every launch goes to the hipStream_t passed as `void* stream`; no implicit
 *     synchronisation, no allocation -> all entry points are hipGraph-capturable.
 *   - layouts are the reference's: feature maps NHWC flattened to [N, P=H*W, C]; 1x1-conv
 *     weights [Cin, Cout] (TF HWIO [1,1,Cin,Cout] squeezed); float32 unless stated.
 *
 * Symbols: N batch, P = H*W pixels, C feature channels (2048), Ca channels of the attention
 * input (C for cfg 002, 768 for cfg 003), K classes, M bottom-up maps (1, or K for per-class),
 * J pose keypoints (16), Cp pose pre-logit channels (768).
 */
#ifndef APA_H_
#define APA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APA_VERSION 300 /* major*10000 + minor*100 + patch */

typedef enum apa_status {
  APA_OK = 0,
  APA_ERR_INVALID_ARG = -1,   /* null pointer, non-positive dimension, bad enum           */
  APA_ERR_UNSUPPORTED = -2,   /* shape/dtype combination the kernels are not built for    */
  APA_ERR_WORKSPACE = -3,     /* ws_bytes smaller than apa_*_workspace_bytes()            */
  APA_ERR_HIP = -4            /* a HIP runtime call / launch failed (see apa_last_error)  */
} apa_status;

/* dtype of the feature-map tensors (X, Xatt, dX, dXatt, TopDown); parameters are always f32 */
#define APA_DTYPE_F32 0
#define APA_DTYPE_BF16 1

/* flags (mirror cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION_*, src/config.py:194-203)      */
#define APA_FLAG_SOFTMAX_ATT 1u /* _SOFTMAX_ATT: spatial softmax of the bottom-up map       */
#define APA_FLAG_RELU_ATT 2u    /* _RELU_ATT                                                 */
#define APA_FLAG_TRAIN 4u       /* is_training: dropout on the top-down input is active      */
#define APA_FLAG_RNG_DEVICE 8u  /* `offset` is the ADDRESS of a uint64 step counter in HBM: the   */
                                /* kernels read it at run time and apa_attn_pool_bwd adds 1 to it */
                                /* when it is done, so a captured hipGraph draws a fresh dropout  */
                                /* mask on every replay                                           */
#define APA_FLAG_RELU_INPUT 16u /* X in memory is the backbone's PRE-activation (block4's residual */
                                /* sum, resnet_v1.py:108-109): the op applies X = max(X, 0) on the  */
                                /* fly in both passes and returns dX already multiplied by [X > 0], */
                                /* so the ReLU's own read+write of the map and its backward pass    */
                                /* (2 + 3 streams of P*C*s bytes) disappear.    */
                                /* M == 1, Xatt == X, C in {1024,2048,4096} (f32) / 2048 (bf16)     */
#define APA_FLAG_DXATT_RANK1 32u /* apa_attn_pool_bwd, M == 1 with a separate attention input (cfg   */
                                /* 003): the gradient w.r.t. Xatt is rank-1, dXatt = dZ (x) Wa; with  */
                                /* this flag `dXatt` is an fp32 [N*P] buffer that receives dZ and the  */
                                /* [N,P,Ca] tensor is never written (its consumer,                     */
                                /* apa_pose_head_bwd_rank1ext, re-forms it in registers)              */

#define APA_FLAG_WS_FROM_FWD 64u /* RESERVED for the library: apa_attn_head_train_step sets it on its own     */
                                /* backward half (same workspace, nothing in between: operands the forward */
                                /* pass prepared there are reused).  The public apa_attn_pool_bwd* entry    */
                                /* points clear it -- a caller cannot vouch for a workspace's history.      */
#define APA_FLAG_RNG_EXTERNAL 128u /* replay a dropout mask drawn ELSEWHERE (the reference's tf.nn.dropout:   */
                                /* binary = floor(keep_prob + uniform), nets_factory.py:143-146,296): `seed` */
                                /* is the ADDRESS of the keep decisions in HBM, bit-packed -- bit (e & 7) of */
                                /* byte e >> 3 belongs to flat element e of the [N,P,C] map (the *_cat entry */
                                /* points: elements N*P*C .. continue into the [N,P,J] extra channels), the   */
                                /* image is ceil(n/8) bytes rounded up to a multiple of 8; `offset` is        */
                                /* ignored; kept elements are scaled by 1/keep_prob.  Pass the same image to  */
                                /* the backward call.  Excludes APA_FLAG_RNG_DEVICE / APA_FLAG_RELU_INPUT.    */
                                /* A replayed mask is read by the library's run-time-loop kernels (any C that */
                                /* is a whole number of 16-byte vectors; per-class maps take the GEMM route), */
                                /* not by the hash-fused streaming kernels: a parity facility, not a fast path */

#define APA_FLAG_WEIGHT_IMAGES 256u /* per-class maps (M == K): the caller vouches that the padded / concatenated   */
                                /* operand images of Wa, ba, Wt, bt in `ws` are CURRENT -- built there by          */
                                /* apa_per_class_weight_images() and rewritten after every weight change (by that  */
                                /* function again, or by the optimiser's own launch, apa_momentum_sgd_step_images) */
                                /* with nothing else using the workspace in between.  The per-step preparation     */
                                /* launch then produces only what changes per step (the dropout decisions), or     */
                                /* does not run at all (evaluation).  Needs 16-byte aligned features (else         */
                                /* APA_ERR_INVALID_ARG); ignored for M == 1.  A stale image is the caller's bug,   */
                                /* like a stale apa_pose_attn_step_io.W1_bf16.                                    */

int apa_version(void);
/* Thread-local, never NULL; describes the last failure on the calling thread. */
const char* apa_last_error(void);
const char* apa_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Attentional pooling, forward.   Replaces nets_factory.py:247-328 + the squeeze at :350-351.
 *
 *   Z = Xatt . Wa + ba                      [N,P,M]     'Conv2d_PrePose_Attn'   (:259-270)
 *   A = f(Z), f = id | softmax over P | relu                                     (:276-287)
 *   Xt = X * mask / keep_prob  (APA_FLAG_TRAIN) else X             slim.dropout  (:296)
 *   T = Xt . Wt + bt                        [N,P,K]     'Conv'                  (:298-309)
 *   logits[n,k] = (1/P) sum_p A[n,p,m(k)] * T[n,p,k],  m(k) = 0 if M==1 else k   (:322-325)
 *
 * X     [N,P,C]   dtype            Xatt  [N,P,Ca]  dtype; pass Xatt == X (Ca == C) for cfg 002
 * Wa    [Ca,M] f32, ba [M] f32     Wt    [C,K] f32, bt [K] f32
 * logits[N,K] f32 (out)            att   [N,P,M] f32 (out) = end_points['PosePrelogitsBasedAttention']
 * zsave (out, saved for backward)  M==1: [N,C] f32 = (1/P) sum_p A[n,p] Xt[n,p,:]
 *                                  M==K: [N,P,K] f32 = the top-down map T
 * abar  [N]   f32 (out, M==1 only) = (1/P) sum_p A[n,p]             -- saved for backward
 * topdown [N,P,K] dtype or NULL    = end_points['TopDownAttention']; only materialised on request
 *                                    (the reference needs it for eval.py --ept dumps only)
 * ws / ws_bytes: scratch of at least apa_attn_pool_workspace_bytes(...) bytes.
 * seed/offset: counter-based dropout RNG key (APA_FLAG_TRAIN); the same pair must be passed to
 *              the backward call.  apa_dropout_mask() materialises the identical mask.  With
 *              APA_FLAG_RNG_EXTERNAL the mask is the caller's own (see the flag).
 * M == 1 runs the factorised HBM-bound path (T is never formed); M == K the dense MFMA path.
 */
size_t apa_attn_pool_workspace_bytes(int N, int P, int C, int Ca, int K, int M, unsigned flags);

int apa_attn_pool_fwd(const void* X, const void* Xatt, const float* Wa, const float* ba,
                      const float* Wt, const float* bt, float* logits, float* att, float* zsave,
                      float* abar, void* topdown, void* ws, size_t ws_bytes, int N, int P, int C,
                      int Ca, int K, int M, unsigned flags, float keep_prob, uint64_t seed,
                      uint64_t offset, int dtype, void* stream);

/* Attentional pooling, backward (the reference has none: TF autodiff of the ops above,
 * model_deploy.py:263).  G = dLoss/dlogits [N,K] f32.  Outputs (all written, not accumulated):
 *   dX [N,P,C] dtype; dXatt [N,P,Ca] dtype or NULL when Xatt == X (its term is folded into dX);
 *   dWa [Ca,M], dba [M], dWt [C,K], dbt [K]  f32.
 * att/zsave/abar are the tensors the forward call produced.
 */
int apa_attn_pool_bwd(const void* X, const void* Xatt, const float* Wa, const float* ba,
                      const float* Wt, const float* bt, const float* att, const float* zsave,
                      const float* abar, const float* G, void* dX, void* dXatt, float* dWa,
                      float* dba, float* dWt, float* dbt, void* ws, size_t ws_bytes, int N, int P,
                      int C, int Ca, int K, int M, unsigned flags, float keep_prob, uint64_t seed,
                      uint64_t offset, int dtype, void* stream);

/* Writes the {0,1} keep-mask that APA_FLAG_TRAIN applies to X (uint8 [N*P*C]); used by the
 * parity tests to hand the oracle the exact mask (TF's own RNG stream is not reproducible). */
int apa_dropout_mask(uint8_t* mask, size_t n_elems, float keep_prob, uint64_t seed, uint64_t offset,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * PoseLogits head.  Replaces nets_factory.py:147-160:
 *   Ppre = relu(X . W1 + b1)   [N,P,Cp]  dtype    'PoseLogits/ExtraConv2d_1x1'  (Cp = 768)
 *   Pl   = Ppre . W2 + b2      [N,P,J]   f32      'PoseLogits/Conv2d_1c_1x1'    (J = 16)
 * W1 [C,Cp], b1 [Cp], W2 [Cp,J], b2 [J] f32.  Dense: bf16 features run on the bf16 MFMA with
 * fp32 accumulation, fp32 features on the exact f32 MFMA.
 * Backward (TF autodiff in the reference):  dPl [N,P,J] f32 = gradient of the pose loss (or NULL);
 * dPpre_ext [N,P,Cp] dtype = gradient arriving at Ppre from the attention branch of cfg 003 (the
 * dXatt of apa_attn_pool_bwd) or NULL.  Outputs dW1 [C,Cp], db1 [Cp], dW2 [Cp,J], db2 [J] f32 and
 * dX [N,P,C] dtype, overwritten or (accumulate_dX & 1) added to what the buffer already holds.
 * accumulate_dX & APA_POSE_WS_FROM_FWD: `ws` is the workspace of the matching apa_pose_head_fwd call and has
 * not been written since -- the bf16 copy of W1 the forward call left there is reused, not rebuilt.
 */
#define APA_POSE_WS_FROM_FWD 2
size_t apa_pose_head_workspace_bytes(int N, int P, int C, int Cp, int J, int dtype);
int apa_pose_head_fwd(const void* X, const float* W1, const float* b1, const float* W2,
                      const float* b2, void* Ppre, float* Pl, void* ws, size_t ws_bytes, int N, int P,
                      int C, int Cp, int J, int dtype, void* stream);
int apa_pose_head_bwd(const void* X, const float* W1, const float* W2, const void* Ppre,
                      const float* dPl, const void* dPpre_ext, void* dX, int accumulate_dX, float* dW1,
                      float* db1, float* dW2, float* db2, void* ws, size_t ws_bytes, int N, int P, int C,
                      int Cp, int J, int dtype, void* stream);
/* Same, with the external gradient in rank-1 form: dPpre_ext[r,j] = ext_row[r] * ext_col[j]
 * (ext_row f32 [N*P] = the dZ that apa_attn_pool_bwd returns under APA_FLAG_DXATT_RANK1, ext_col f32
 * [Cp] = the attention weights Wa): one write and one read of an [N,P,Cp] tensor less per step. */
int apa_pose_head_bwd_rank1ext(const void* X, const float* W1, const float* W2, const void* Ppre,
                               const float* dPl, const float* ext_row, const float* ext_col, void* dX,
                               int accumulate_dX, float* dW1, float* db1, float* dW2, float* db2,
                               void* ws, size_t ws_bytes, int N, int P, int C, int Cp, int J, int dtype,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * Action loss: tf.losses.softmax_cross_entropy(one_hot(labels,K), logits, weights=wt)
 * (src/loss.py:74-80) fused with its gradient.
 *   loss[0] = wt/N * sum_n -log_softmax(logits[n])[labels[n]];  G = wt/N * (softmax - onehot)
 * `grad_scale` multiplies G only (1/num_clones of model_deploy.py:223-225 goes here).
 * labels int64 [N]; loss f32 [1+N]: loss[0] = weighted batch-mean loss, loss[1+n] = unweighted
 * per-example cross-entropy; G f32 [N,K] or NULL; probs f32 [N,K] or NULL (eval.py:196);
 * pred int64 [N] or NULL (argmax, first maximal index -- eval.py:193).
 */
int apa_softmax_xent_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* G,
                             float* probs, int64_t* pred, int N, int K, float wt, float grad_scale,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * One training step of the head as ONE host call: apa_attn_pool_fwd -> apa_softmax_xent_fwd_bwd ->
 * apa_attn_pool_bwd with the same arguments, on the same stream, launching the same kernels (the
 * three entry points above are called back to back; results are bit-identical to calling them
 * separately).  The reference runs this sequence inside one `sess.run(train_op)` (src/train.py:
 * 529-566) in TensorFlow's C++ executor, with no per-op interpreter cost; at ~55 us per step on
 * MI355X three separately marshalled foreign calls would make the HOST the bottleneck, so the
 * native sequence is part of the boundary.  All buffers are caller-owned (see the three functions
 * for shapes); loss f32 [1+N], G f32 [N,K].
 * Every output -- logits, loss and G included -- is valid only when the WHOLE call returns APA_OK: inside the
 * call launches are shared between the three ops (for the per-class maps with K <= 64 the cross-entropy is taken
 * by the first backward kernel, which is what writes logits / loss / G), so an error return from the second half
 * (e.g. APA_ERR_WORKSPACE) leaves them undefined.
 */
int apa_attn_head_train_step(const void* X, const void* Xatt, const float* Wa, const float* ba,
                             const float* Wt, const float* bt, const int64_t* labels, float loss_wt,
                             float grad_scale, float* logits, float* att, float* zsave, float* abar,
                             float* loss, float* G, void* dX, void* dXatt, float* dWa, float* dba,
                             float* dWt, float* dbt, void* ws, size_t ws_bytes, int N, int P, int C,
                             int Ca, int K, int M, unsigned flags, float keep_prob, uint64_t seed,
                             uint64_t offset, int dtype, void* stream);

/* One evaluation step of the head as ONE host call (eval.py:181-197: network_fn, then
 * tf.argmax(logits,1) and tf.nn.softmax(logits,-1) in the same sess.run): apa_attn_pool_fwd followed
 * by apa_softmax_xent_fwd_bwd without the gradient.  probs f32 [N,K]; pred int64 [N] (first maximal
 * index); labels / loss ([1+N], see above) may both be NULL when no ground truth is at hand.  */
int apa_attn_head_eval_step(const void* X, const void* Xatt, const float* Wa, const float* ba,
                            const float* Wt, const float* bt, const int64_t* labels, float* logits,
                            float* att, float* zsave, float* abar, float* loss, float* probs,
                            int64_t* pred, void* ws, size_t ws_bytes, int N, int P, int C, int Ca, int K,
                            int M, unsigned flags, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * One host call for the whole cfg 003 head step (003_MPII_ResNet_withPoseAttention.yaml): what the reference
 * runs as one sess.run of   PoseLogits head (nets_factory.py:147-160) -> attention from pose_pre_logits
 * (:247-270, M = 1) -> dropout + top-down conv + attention-weighted mean (:296-328) -> pose L2 +
 * softmax cross-entropy (src/loss.py:11-80) -> tf.gradients of both w.r.t. every head variable and conv5.
 * Same results as the sequence apa_pose_head_fwd, apa_pose_l2_loss_fwd_bwd, apa_attn_head_train_step
 * (APA_FLAG_DXATT_RANK1), apa_pose_head_bwd_rank1ext -- which is what it runs when the fused kernels do not
 * serve the shape -- but inside one call neighbouring ops share launches (bf16 features, J = 16,
 * Cp in {256,512,768,1024}, K <= 512): the Pl product also emits the attention logits and the pose loss
 * gradient; the pose head's backward rows pass also emits dWa / dba; its column-sum launch also finishes the
 * pose loss and advances the dropout counter: 18 -> 13 launches at the benchmark shape.
 * io->W1_bf16 (optional): a bf16 copy of W1 [C,Cp] the caller keeps current (apa_momentum_sgd_step_shadow
 * rewrites it in the optimizer's own launch); NULL = converted inside the call.
 * io->W2T_bf16 (optional, round 5): W2 TRANSPOSED and rounded to bf16, [16][Cp + 16] (rows J..15 and the 16 pad
 * columns zero), 16-byte aligned, kept current the same way (apa_momentum_sgd_step_images with the map "element
 * (c, j) of W2 -> dst[j * (Cp + 16) + c]": a = 1, d = Cp + 16, c_shift = b = e = 0): the Pl product copies this
 * 24.5 KB image verbatim into LDS (LDS-DMA) instead of converting and transposing the fp32 W2 in every block;
 * NULL = as before.  Same values.
 * All pointers device memory; X/Ppre/dX of `dtype`, everything else f32 (labels int64, pose_valid uint8).
 * flags: APA_FLAG_SOFTMAX_ATT / RELU_ATT / TRAIN / RNG_DEVICE as for apa_attn_pool_fwd.
 */
typedef struct apa_pose_attn_step_io {
  const void* X;            /* [N,P,C]                                                   */
  const float* W1;          /* [C,Cp]   PoseLogits/ExtraConv2d_1x1                       */
  const float* b1;          /* [Cp]                                                      */
  const float* W2;          /* [Cp,J]   PoseLogits/Conv2d_1c_1x1                         */
  const float* b2;          /* [J]                                                       */
  const void* W1_bf16;      /* optional bf16 [C,Cp] copy of W1, or NULL                  */
  const void* W2T_bf16;     /* optional bf16 [16,Cp+16] transposed copy of W2, or NULL   */
  const float* Wa;          /* [Cp,1]   Conv2d_PrePose_Attn                              */
  const float* ba;          /* [1]                                                       */
  const float* Wt;          /* [C,K]    top-down conv                                    */
  const float* bt;          /* [K]                                                       */
  const int64_t* labels;    /* [N]                                                       */
  const float* pose_labels; /* [N,P,J]                                                   */
  const uint8_t* pose_valid;/* [N,J]                                                     */
  float action_wt, pose_wt, grad_scale;
  /* activations (outputs) */
  void* Ppre;               /* [N,P,Cp] dtype                                            */
  float* Pl;                /* [N,P,J]                                                   */
  float* att;               /* [N,P]                                                     */
  float* logits;            /* [N,K]                                                     */
  float* zsave;             /* [N,C]                                                     */
  float* abar;              /* [N]                                                       */
  float* loss_action;       /* [1+N]  (as apa_attn_head_train_step)                      */
  float* loss_pose;         /* [1]                                                       */
  /* gradients (outputs) */
  float* G;                 /* [N,K]   d loss / d logits                                 */
  float* dPl;               /* [N,P,J]                                                   */
  float* dZ;                /* [N*P]   gradient at the attention logits (rank-1 factor)  */
  void* dX;                 /* [N,P,C] dtype                                             */
  float *dW1, *db1, *dW2, *db2, *dWa, *dba, *dWt, *dbt;
  /* scratch */
  void* ws_pool;  size_t ws_pool_bytes;   /* apa_attn_pool_workspace_bytes(N,P,C,Cp,K,1,flags) */
  void* ws_pose;  size_t ws_pose_bytes;   /* apa_pose_head_workspace_bytes(N,P,C,Cp,J,dtype)   */
} apa_pose_attn_step_io;
int apa_pose_attn_train_step(const apa_pose_attn_step_io* io, int N, int P, int C, int Cp, int J, int K,
                             unsigned flags, float keep_prob, uint64_t seed, uint64_t offset, int dtype,
                             void* stream);

/* Pose loss: src/loss.py:29-70 ('l2', LOSS_FN_POSE_SAMPLED off) fused with its gradient.
 *   loss[0] = wt * sum_j mean_n( valid[n,j] ? 0.5*sum_p (Pl-lbl)^2 / (N*P) : 0 )
 *   dPl = grad_scale * wt * valid[n,j] * (Pl - lbl) / (N*N*P)
 * Pl, lbl [N,P,J] f32; valid uint8 [N,J]; dPl [N,P,J] f32 or NULL; loss f32 [1];
 * ws: at least apa_pose_l2_workspace_bytes(N,P,J) bytes of device scratch. */
size_t apa_pose_l2_workspace_bytes(int N, int P, int J);
int apa_pose_l2_loss_fwd_bwd(const float* Pl, const float* lbl, const uint8_t* valid, float* loss,
                             float* dPl, void* ws, size_t ws_bytes, int N, int P, int J, float wt,
                             float grad_scale, void* stream);

/* The other action losses gen_losses accepts (src/loss.py:81-101), value + gradient in one launch:
 *   APA_ACTION_LOSS_L2             tf.losses.mean_squared_error(one_hot(labels,K), logits, weights=wt):
 *                                  loss = wt * mean_{n,k} (logits - onehot)^2;  labels int64 [N]
 *   APA_ACTION_LOSS_MULTI_LABEL    mean(tf.nn.weighted_cross_entropy_with_logits(targets, logits,
 *                                  pos_weight)) added with tf.losses.add_loss -- action_loss_wt is NOT
 *                                  applied by the reference (loss.py:93-97), so `wt` is ignored;
 *                                  labels f32 [N,K] multi-hot; the reference passes pos_weight = 10
 *   APA_ACTION_LOSS_MULTI_LABEL_2  tf.losses.sigmoid_cross_entropy(labels, logits) * wt; labels f32 [N,K]
 * loss f32 [1]; G f32 [N,K] or NULL = grad_scale * dloss/dlogits. */
#define APA_ACTION_LOSS_L2 1
#define APA_ACTION_LOSS_MULTI_LABEL 2
#define APA_ACTION_LOSS_MULTI_LABEL_2 3
int apa_action_loss_fwd_bwd(int kind, const float* logits, const void* labels, float* loss, float* G,
                            int N, int K, float wt, float grad_scale, float pos_weight, void* stream);

/* Pose loss with cfg.TRAIN.LOSS_FN_POSE_SAMPLED (src/loss.py:36-52), literally -- including the
 * loop header's swapped names, by which "keep all positive pixels" tests the LOGITS:
 *   ratio_j = #(lbl[..,j] > 0) / (N*P);  sel = (uniform < ratio_j) * (lbl == 0)
 *   mask = (sel + (Pl > 0)) > 0;         loss = wt * sum_j mean_n( valid ? 0.5 * mean_p (mask*(Pl-lbl))^2 : 0 )
 * uniform f32 [N,P,J]: the caller's tf.random_uniform draws in [0,1) (passed in like a dropout mask:
 * TF's stream cannot be reproduced); mask_out f32 [N,P,J] or NULL = end_points['PoseLossMask'];
 * dPl [N,P,J] or NULL (the mask is a constant of the differentiation, as in TF). */
int apa_pose_sampled_loss_fwd_bwd(const float* Pl, const float* lbl, const uint8_t* valid,
                                  const float* uniform, float* loss, float* dPl, float* mask_out, int N,
                                  int P, int J, float wt, float grad_scale, void* stream);

/* tf.image.resize_images(BILINEAR) as of TF 1.1 (legacy rule: src = dst * in/out, no half-pixel
 * offset; src/loss.py:14-22 on the label map, device tensors): in f32 [N,h,w,C] -> out [N,oh,ow,C]. */
int apa_resize_bilinear_tf1(const float* in, float* out, int N, int h, int w, int C, int out_h, int out_w,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Label generator: PoseToHeatmapOp::Compute, src/custom_ops/pose_to_heatmap.cc:35-96.
 * HOST function (the reference op is DEVICE_CPU and runs inside the input pipeline).
 * pose_host int64 [n_vals], n_vals % (3*out_channels) == 0, triples (x, y, is_visible);
 * heatmap_host f32 [out_ht, out_wd, out_channels] with out_ht = (int)(im_ht*out_wd*1.0/im_wd)
 * (query it with apa_pose_to_heatmap_out_ht); valid_host uint8 [out_channels].
 */
int64_t apa_pose_to_heatmap_out_ht(int64_t im_ht, int64_t im_wd, int64_t out_wd);
int apa_pose_to_heatmap(const int64_t* pose_host, int64_t n_vals, int64_t im_ht, int64_t im_wd,
                        int64_t out_wd, int out_channels, float marker_wd_ratio, int do_gauss_blur,
                        float* heatmap_host, uint8_t* valid_host);

/* Label post-processing: _replay_augmentation (src/preprocess_pipeline.py:21-45) + :195-214.
 * HOST function.  hm_host uint8 [h,w,J] (the python wrapper's *255 heat-map); the crop
 * (crop_y, crop_x, crop_h, crop_w) is given in the coordinates of the orig_h x orig_w image it was
 * recorded on and is rescaled to the heat-map with the reference's float32 ratios and truncation;
 * flip != 0 mirrors left-right after the crop; values become float (x/255), are min-max
 * normalised ((x - min) / (max(x - min) + eps), eps = cfg.EPS = 1e-14) and resized with the TF1
 * legacy bilinear rule to out_side x out_side.  out_host f32 [out_side, out_side, J]. */
int apa_pose_label_replay_resize(const uint8_t* hm_host, int h, int w, int J, int orig_h, int orig_w,
                                 int crop_y, int crop_x, int crop_h, int crop_w, int flip,
                                 int out_side, float eps, float* out_host);

/* The same label path on the DEVICE for a whole batch, canvas-free (training call: no blur,
 * src/preprocess_pipeline.py:162): apa_pose_to_heatmap -> *255 -> apa_pose_label_replay_resize fused,
 * one block per image; bit-identical to the two host functions above (eps: any value << 1, e.g.
 * cfg.EPS = 1e-14, gives the same result on a binary canvas).
 *   pose    int64 [N][max_vals]  (x, y, is_visible) triples, n_vals[n] of them used (multiple of 3*J,
 *           at most 32 people)
 *   geom    int32 [N][9] = im_ht, im_wd, aug_ht, aug_wd, crop_y, crop_x, crop_h, crop_w, flip.
 *           TWO frames, as in the reference: the keypoints are scaled onto the canvas with the size
 *           of the ORIGINAL image (im_ht x im_wd, preprocess_pipeline.py:155-157), while the crop
 *           was recorded on the image AFTER the aspect-preserving resize to RESIZE_SIDE (aug_ht x
 *           aug_wd = preproc_info['image_shape'], vgg_preprocessing.py:325) and is rescaled to the
 *           canvas with ratio = canvas / aug size (_replay_augmentation, :29-36) -- the orig_h x
 *           orig_w of apa_pose_label_replay_resize
 *   labels  f32 [N, out_side, out_side, J];  valid uint8 [N, J];  status int32 [N]: 0 = ok, 1 = the
 *           host path would have failed for this image (bad crop / sizes): its label is all zero. */
int apa_pose_labels_device(const int64_t* pose, const int32_t* n_vals, const int32_t* geom, int N,
                           int max_vals, int out_wd, int J, float marker_wd_ratio, int out_side,
                           float* labels, uint8_t* valid, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * Video frame pooling of per-frame logits, nets_factory.py:354-374.  logits f32 [B*F, K] (F
 * consecutive rows per video) -> pooled f32 [B, K].
 *   w == NULL : pooled = mean over the F frames                                   (:374)
 *   w != NULL : temporal attention (cfg.NET.USE_TEMPORAL_ATT, :362-372): tatt[b,f] =
 *               logits[b,f,:] . w + b[0]; pooled = mean_f(logits * tatt); tatt f32 [B*F] is an
 *               output (end_points['TemporalAttention']) and is needed by the backward call.
 * Backward: dpooled [B,K] -> dlogits [B*F,K], dw [K], db [1]; dlda_ws: B*F floats of scratch. */
int apa_frame_pool_fwd(const float* logits, const float* w, const float* b, float* pooled, float* tatt,
                       int B, int F, int K, void* stream);
int apa_frame_pool_bwd(const float* logits, const float* w, const float* tatt, const float* dpooled,
                       float* dlogits, float* dw, float* db, float* dlda_ws, int B, int F, int K,
                       void* stream);

/* Backward of the global average pool of the attention-free head (cfg 001, resnet_v1.py:206-208
 * `tf.reduce_mean(net, [1, 2])`):  dX[n,p,c] = dz[n,c] / P.   dz f32 [N,C]; dX dtype [N,P,C].
 * (The forward pool is apa_attn_pool_fwd with a constant attention map: its zsave output.) */
int apa_spatial_mean_bwd(const float* dz, void* dX, int N, int P, int C, int dtype, void* stream);

/* zero_out_channels.cc:18-51:  out[n,h,w,c] = channels[c] ? in[n,h,w,c] : 0   (f32, device). */
int apa_zero_out_channels(const float* in, const uint8_t* channels, float* out, size_t n_outer,
                          int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-call hooks (communication / compute overlap, measurement).  The library keeps NO state
 * between calls and none per thread: a caller that wants one of the aids below passes this struct
 * to the *_ex form of an entry point; NULL (or the plain entry points) means "no hooks".  Every
 * member is a hipEvent_t created by the caller (or NULL).  Purely ordering / timing aids: results
 * are unaffected.
 *
 *   grad_ready_event        recorded on the call's stream as soon as the classifier gradients
 *                           dWt / dbt -- 99.7 % of the all-reduce payload -- are final, i.e. after the
 *                           FIRST kernel of the backward pass, before the streaming pass starts: a
 *                           data-parallel trainer starts the RCCL all-reduce of that part of the
 *                           bucket on a second stream while dX / dWa are still being produced
 *                           (deploy.OverlappedGradientSum, bench.py --gpus N).
 *   td_weights_ready_event  the mirror image on the forward side: the stream waits for it
 *                           (hipStreamWaitEvent) immediately before the first kernel that reads
 *                           td_weights / td_biases -- on the M == 1 path that is the logits product,
 *                           AFTER the pooling pass, which only needs the attention weights.  The
 *                           trainer records it on its communication stream once the all-reduce (and
 *                           optimizer update) of td_weights of the previous step is done, so that
 *                           collective hides under the streaming backward pass of step k AND the
 *                           pooling pass of step k+1.
 *   prof_fwd_start/stop     the M == 1 streaming kernels (m1s_pool_fwd_kernel / m1s_bwd_main_kernel) are
 *   prof_bwd_start/stop     launched with hipExtLaunchKernel(start, stop), which brackets exactly that
 *                           dispatch on its stream; hipEventElapsedTime(start, stop) measured ~1.2 us
 *                           above rocprofv3's begin -> end of the same kernel on MI355X (profiles/), a
 *                           constant the caller may keep in mind but bench.py does not subtract.
 */
typedef struct apa_hooks {
  void* grad_ready_event;
  void* td_weights_ready_event;
  void* prof_fwd_start;
  void* prof_fwd_stop;
  void* prof_bwd_start;
  void* prof_bwd_stop;
} apa_hooks;

int apa_attn_pool_fwd_ex(const apa_hooks* hooks, const void* X, const void* Xatt, const float* Wa,
                         const float* ba, const float* Wt, const float* bt, float* logits, float* att,
                         float* zsave, float* abar, void* topdown, void* ws, size_t ws_bytes, int N,
                         int P, int C, int Ca, int K, int M, unsigned flags, float keep_prob,
                         uint64_t seed, uint64_t offset, int dtype, void* stream);
int apa_attn_pool_bwd_ex(const apa_hooks* hooks, const void* X, const void* Xatt, const float* Wa,
                         const float* ba, const float* Wt, const float* bt, const float* att,
                         const float* zsave, const float* abar, const float* G, void* dX, void* dXatt,
                         float* dWa, float* dba, float* dWt, float* dbt, void* ws, size_t ws_bytes,
                         int N, int P, int C, int Ca, int K, int M, unsigned flags, float keep_prob,
                         uint64_t seed, uint64_t offset, int dtype, void* stream);
int apa_attn_head_train_step_ex(const apa_hooks* hooks, const void* X, const void* Xatt,
                                const float* Wa, const float* ba, const float* Wt, const float* bt,
                                const int64_t* labels, float loss_wt, float grad_scale, float* logits,
                                float* att, float* zsave, float* abar, float* loss, float* G, void* dX,
                                void* dXatt, float* dWa, float* dba, float* dWt, float* dbt, void* ws,
                                size_t ws_bytes, int N, int P, int C, int Ca, int K, int M,
                                unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                                int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * ..._WITH_POSE_FEAT (nets_factory.py:289-295): `last_conv = tf.concat([last_conv, pose_logits], -1)`
 * before the dropout and the top-down conv, i.e. J extra fp32 channels Xext [N,P,J] (the PoseLogits
 * output, after the optional ..._2LAYER conv) next to the C channels of X.  The concatenated tensor is
 * never formed: the *_cat entry points take the two parts separately.  With `cat` given
 *   Wt / dWt are the full td_weights [C+J, K] (rows C.. belong to the extra channels),
 *   the dropout mask of the extra channels continues X's flat element stream at index N*P*C
 *   (apa_dropout_mask over N*P*C + N*P*J elements returns both parts),
 *   zext [N,J] is written by the forward call and must be passed unchanged to the backward call,
 *   dXext [N,P,J] receives the gradient w.r.t. Xext (backward call only).
 * M == 1 only, no APA_FLAG_RELU_INPUT, no TopDownAttention dump; cat == NULL is the plain call.
 */
typedef struct apa_concat_feat {
  const float* Xext;
  int J;
  float* zext;
  float* dXext;
} apa_concat_feat;

int apa_attn_pool_fwd_cat(const apa_concat_feat* cat, const apa_hooks* hooks, const void* X,
                          const void* Xatt, const float* Wa, const float* ba, const float* Wt,
                          const float* bt, float* logits, float* att, float* zsave, float* abar,
                          void* topdown, void* ws, size_t ws_bytes, int N, int P, int C, int Ca, int K,
                          int M, unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                          int dtype, void* stream);
int apa_attn_pool_bwd_cat(const apa_concat_feat* cat, const apa_hooks* hooks, const void* X,
                          const void* Xatt, const float* Wa, const float* ba, const float* Wt,
                          const float* bt, const float* att, const float* zsave, const float* abar,
                          const float* G, void* dX, void* dXatt, float* dWa, float* dba, float* dWt,
                          float* dbt, void* ws, size_t ws_bytes, int N, int P, int C, int Ca, int K,
                          int M, unsigned flags, float keep_prob, uint64_t seed, uint64_t offset,
                          int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused optimizer step (src/train.py:90-94 tf.train.MomentumOptimizer + the slim L2 regulariser of
 * models/slim/nets/resnet_utils.py:241 on conv weights):  for every parameter segment i
 *     g   = grad_scale * grad_flat[off_i ...] + weight_decay[i] * w_i
 *     acc = momentum * acc + g ;   w_i -= lr * acc
 * weights[i]: device pointers (fp32) to the nseg parameter tensors, sizes[i] their element counts
 * (host arrays); grad_flat / acc_flat: the flat all-reduce bucket and the momentum accumulators in
 * the same concatenated order.  weight_decay[i] is 0 for biases.  One launch, any alignment.
 */
#define APA_SGD_MAX_SEGMENTS 16
int apa_momentum_sgd_step(int nseg, float* const* weights, const size_t* sizes,
                          const float* weight_decay, const float* grad_flat, float* acc_flat,
                          float lr, float momentum, float grad_scale, void* stream);
/* The same update; bf16_shadow[i] (HOST array of nseg device pointers, entries may be NULL) additionally receives
 * the UPDATED weights of segment i rounded to bf16 (round-to-nearest-even) -- the operand copy the bf16 MFMA
 * products read (apa_pose_attn_step_io.W1_bf16), kept current by the optimizer's own launch instead of a
 * conversion kernel in every step. */
int apa_momentum_sgd_step_shadow(int nseg, float* const* weights, const size_t* sizes,
                                 const float* weight_decay, const float* grad_flat, float* acc_flat,
                                 float lr, float momentum, float grad_scale, void* const* bf16_shadow,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight images of the per-class head (M == K) kept current at WEIGHT-UPDATE time instead of being rebuilt by
 * every step.  The per-class products read Wa | Wt (and ba | bt) as padded, concatenated bf16 images inside the
 * workspace (K <= 64: `[C/64][128][64]` k-tile-major for the forward product and `[C][Wt | Wa]` for dX, biases as
 * one f32 [128] row; any K: `[C][Wt (Kp) | Wa (Kp)]`, ba padded to f32 [Kp]); without APA_FLAG_WEIGHT_IMAGES each
 * forward / train-step call rebuilds them from the fp32 parameters (5-6 us of a 58 us HMDB-51 step).
 *
 * apa_per_class_weight_images: builds every image the shape (N,P,C,Ca,K,dtype) can need in `ws` (>= apa_attn_pool_
 *   workspace_bytes(..., M = K)) -- one or two launches, for initialisation and after a checkpoint restore -- and, when
 *   `maps` is given, describes them: maps[i] says "element (c, k) of parameter `role` goes to
 *   dst[(c >> c_shift) * a + (c & ((1 << c_shift) - 1)) * b + k * d + e]" (bf16, or f32 when is_f32).  *nmaps
 *   receives the count (<= APA_WIMG_MAX; 0: this shape keeps no images and the flag changes nothing).
 * apa_momentum_sgd_step_images: apa_momentum_sgd_step_shadow that ALSO rewrites those images from the updated
 *   weights in the same launch: images[i] belongs to parameter segment image_segment[i].
 */
#define APA_WIMG_ROLE_WA 0
#define APA_WIMG_ROLE_BA 1
#define APA_WIMG_ROLE_WT 2
#define APA_WIMG_ROLE_BT 3
#define APA_WIMG_MAX 12
#define APA_WIMG_PER_SEGMENT 3
typedef struct apa_weight_image {
  void* dst;          /* device pointer INTO the workspace                                   */
  int role;           /* APA_WIMG_ROLE_*: the parameter this image is made from              */
  int is_f32;         /* destination element type: 0 = bf16 (round to nearest even), 1 = f32 */
  int cols;           /* columns of the parameter matrix: flat element i is (c, k) = (i / cols, i % cols) */
  int c_shift;
  int a, b, d, e;
} apa_weight_image;
int apa_per_class_weight_images(const float* Wa, const float* ba, const float* Wt, const float* bt, void* ws,
                                size_t ws_bytes, int N, int P, int C, int Ca, int K, int dtype,
                                apa_weight_image* maps, int* nmaps, void* stream);
int apa_momentum_sgd_step_images(int nseg, float* const* weights, const size_t* sizes, const float* weight_decay,
                                 const float* grad_flat, float* acc_flat, float lr, float momentum,
                                 float grad_scale, void* const* bf16_shadow, const apa_weight_image* images,
                                 const int* image_segment, int nimages, void* stream);

/* The other two optimisers src/train.py:84-100 can select (TRAIN.OPTIMIZER 'adam' / 'rmsprop'; no shipped YAML
 * does), same flat layout, two slot buffers, one launch, optional bf16 shadows (may be NULL) as above:
 *   apa_adam_step     tf.train.AdamOptimizer(lr, beta1, beta2, epsilon):  m += (g - m)(1 - beta1);
 *                     v += (g^2 - v)(1 - beta2);  w -= lr_t m / (sqrt(v) + epsilon), where the CALLER passes
 *                     lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t) for update number t = 1, 2, ...; slots start at 0.
 *   apa_rmsprop_step  tf.train.RMSPropOptimizer(lr, decay, momentum, epsilon):  ms += (g^2 - ms)(1 - decay);
 *                     mom = momentum mom + lr g / sqrt(ms + epsilon);  w -= mom;  `ms` starts at ONE, `mom` at 0.
 * g = grad_scale * grad_flat[...] + weight_decay[i] * w_i in both. */
int apa_adam_step(int nseg, float* const* weights, const size_t* sizes, const float* weight_decay,
                  const float* grad_flat, float* m_flat, float* v_flat, float lr_t, float beta1, float beta2,
                  float epsilon, float grad_scale, void* const* bf16_shadow, void* stream);
int apa_rmsprop_step(int nseg, float* const* weights, const size_t* sizes, const float* weight_decay,
                     const float* grad_flat, float* ms_flat, float* mom_flat, float lr, float decay, float momentum,
                     float epsilon, float grad_scale, void* const* bf16_shadow, void* stream);

/* ------------------------------------------------------------------------------------------
 * TRAIN.ITER_SIZE accumulation (src/train.py:529-566: `ref = grad` on the first micro-step, `ref += grad` on the
 * following ones, `apply_gradients(ref / ITER_SIZE)` on the last) for micro-batches whose gradients were written
 * into separate flat buckets (deploy.OverlappedMicroBatches runs them on separate streams):
 *     out[i] = ((parts[0][i] + parts[1][i]) + ...) * scale           same summation order as the reference
 * parts: HOST array of nparts device pointers (f32 [n] each; out may alias parts[0]); one launch, any alignment.
 */
#define APA_ACC_MAX_PARTS 8
int apa_accumulate_gradients(float* out, const float* const* parts, int nparts, size_t n, float scale,
                             void* stream);
/* The same sum DIVIDED by `divisor` (IEEE fp32 division), i.e. literally `ref_grad / float(ITER_SIZE)`
 * (src/train.py:560-563): bit-identical to the reference's arithmetic for every ITER_SIZE, where the scale form
 * differs by an ulp unless ITER_SIZE is a power of two.  divisor == 0 is rejected. */
int apa_accumulate_gradients_div(float* out, const float* const* parts, int nparts, size_t n, float divisor,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement helpers (bench.py): HIP events owned by the library's HIP runtime, for the prof_*
 * members of apa_hooks.
 */
int apa_prof_event_create(void** event);
int apa_prof_event_destroy(void* event);
int apa_prof_event_record(void* event, void* stream);
int apa_prof_event_elapsed_ms(void* start, void* stop, float* ms); /* both must have completed */

#ifdef __cplusplus
}
#endif
#endif /* APA_H_ */
