"""CPU oracle for the attentional-pooling hot path  --  TEST INFRASTRUCTURE ONLY.

    PARITY STATUS: PINNED TO THE REFERENCE'S OWN GRAPH CODE, not to TensorFlow's kernels.
    The reference's head and losses are plain Python graph construction
    (models/slim/nets/nets_factory.py:94-380, src/loss.py:4-105); the arithmetic of each op lives
    in un-vendored TensorFlow 1.1.0-rc2, which -- like Python 2 and OpenCV -- is absent from this
    image.  tests/golden/make_head_reference.py therefore EXECUTES those two reference files (plus
    src/config.py, the shipped experiments/*.yaml and the backbones' arg_scope functions) from the
    reference tree behind a float64 stand-in for the ~40 TF/slim symbols they call
    (tests/golden/tf1_shim.py, written from TF's documented op semantics, independent of this file)
    and commits the results as tests/golden/ref_head_*.npz / ref_losses.npz: 39 head configurations
    (cfg 002 / 003 from their YAML, softmax / relu / per-class / rank 2-3 / _WITH_POSE_FEAT(+_2LAYER)
    under four arg-scopes / video frame pooling / temporal attention / separate pose tap, training
    mode with the recorded dropout mask) and 12 gen_losses cases.
    tests/test_reference_fixtures_cpu.py holds THIS restatement to those fixtures at 1e-12 --
    logits, end points, every loss and regularisation term, every gradient.
    What remains unpinned (needs TensorFlow itself): float32 rounding of TF's conv / softmax / reduce
    kernels, TF's RNG streams (random draws are a recorded stream here), and cv::circle for the label
    generator (oracle/labels_eval_oracle.py).  The older fixtures tests/golden/attn_*.npz, losses.npz
    are outputs of THIS file (tests/golden/make_golden.py): regression pins only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped package (attentionalpoolingaction_amd/) never imports it.

All citations are relative to /root/reference/.  Tensors are NHWC like the reference.
Every function works in the dtype of its inputs (float32 to mimic TF1, float64 for a
high-precision target) and is differentiable through torch autograd, which plays the role of
TF's `optimizer.compute_gradients` (models/slim/deployment/model_deploy.py:263).

The oracle deliberately follows the *literal* reference formulation -- it materialises the
[N,H,W,K] top-down tensor and multiplies it by the attention map exactly like
nets_factory.py:298-325 -- and NOT the factorised form the HIP kernels use.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------------------
# slim.conv2d(x, Cout, [1,1], activation_fn=None, normalizer_fn=None)
#   == x . W + b   with W stored [1,1,Cin,Cout]  
# --------------------------------------------------------------------------------------
def conv1x1(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """1x1 SAME stride-1 convolution + BiasAdd on an NHWC tensor.

    x: [N,H,W,Cin]; w: [Cin,Cout] (the [1,1,Cin,Cout] HWIO kernel squeezed); b: [Cout].
    Reference call sites: models/slim/nets/nets_factory.py:151-159, 263-269, 300-303.
    """
    y = torch.matmul(x, w)
    if b is not None:
        y = y + b
    return y


def dropout(x: torch.Tensor, keep_prob: float, mask: Optional[torch.Tensor],
            is_training: bool) -> torch.Tensor:
    """slim.dropout -> tf.nn.dropout(x, keep): y = x / keep * mask, identity at eval.

    Reference: nets_factory.py:143-146 (keep = 0.2 when cfg.NET.DROPOUT < 0) and :296.
    TF's RNG stream cannot be reproduced, so the binary keep-mask is an explicit input
    (mask = floor(keep + U[0,1)) in TF; here any {0,1} tensor shaped like x).
    """
    if not is_training:
        return x
    assert mask is not None, "training-mode dropout needs an explicit mask"
    return x / keep_prob * mask.to(x.dtype)


# --------------------------------------------------------------------------------------
# PoseLogits head  -- nets_factory.py:147-160
# --------------------------------------------------------------------------------------
def pose_logits_head(last_conv_pose: torch.Tensor,
                     w1: torch.Tensor, b1: torch.Tensor,
                     w2: torch.Tensor, b2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """pose_pre_logits = relu(conv1x1(X; [C,768])+b1); pose_logits = conv1x1(.; [768,J])+b2.

    Scopes 'PoseLogits/ExtraConv2d_1x1' (:151-157) and 'PoseLogits/Conv2d_1c_1x1' (:158-159);
    no batch-norm (normalizer_fn=None).  Returns (pose_pre_logits, pose_logits).
    """
    pre = torch.relu(conv1x1(last_conv_pose, w1, b1))
    logits = conv1x1(pre, w2, b2)
    return pre, logits


# --------------------------------------------------------------------------------------
# Attentional pooling  -- nets_factory.py:242-328 (+ squeeze :350-352)
# --------------------------------------------------------------------------------------
@dataclass
class AttnFlags:
    """Mirror of the cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION_* flags (src/config.py:182-209)."""
    single_layer_att: bool = True      # _SINGLE_LAYER_ATT : attention from last_conv (cfg 002)
    softmax_att: bool = False          # _SOFTMAX_ATT      : spatial softmax (:276-284)
    relu_att: bool = False             # _RELU_ATT         : relu on the map (:285-286)
    per_class: bool = False            # _PER_CLASS        : nMaps = num_classes (:257)
    rank: int = 1                      # _RANK             (:258, :298)
    with_pose_feat: bool = False       # _WITH_POSE_FEAT   (:289-295)
    with_pose_feat_2layer: bool = False


def spatial_softmax_nhwc(a: torch.Tensor) -> torch.Tensor:
    """nets_factory.py:276-284: NHWC -> NCHW, reshape [N,M,H*W], tf.nn.softmax (last axis,
    max-subtracted), reshape and transpose back.  Works on [N,H,W,M] and [N,H,W,M,R]
    (the reference transposes with perm [0,3,1,2], which only type-checks for rank==1; for
    R>1 we apply the same per-(n,map,rank) spatial softmax, the only sensible reading)."""
    n, h, w = a.shape[0], a.shape[1], a.shape[2]
    rest = a.shape[3:]
    flat = a.reshape(n, h * w, -1).transpose(1, 2)          # [N, M(*R), P]
    flat = torch.softmax(flat, dim=-1)
    return flat.transpose(1, 2).reshape(n, h, w, *rest)


def attentional_pooling(last_conv: torch.Tensor,
                        pose_pre_logits: Optional[torch.Tensor],
                        pose_logits: Optional[torch.Tensor],
                        att_w: Sequence[torch.Tensor], att_b: Sequence[torch.Tensor],
                        td_w: Sequence[torch.Tensor], td_b: Sequence[torch.Tensor],
                        flags: AttnFlags,
                        is_training: bool = False,
                        keep_prob: float = 0.2,
                        dropout_mask: Optional[torch.Tensor] = None,
                        pose_feat_w: Optional[torch.Tensor] = None,
                        pose_feat_bn: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                        pose_feat_b: Optional[torch.Tensor] = None,
                        arg_scope: str = 'resnet',
                        ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Literal restatement of the `USE_POSE_PRELOGITS_BASED_ATTENTION` branch.

    last_conv       [N,H,W,C]   backbone tap (nets_factory.py:136-140)
    pose_pre_logits [N,H,W,768] only read when not flags.single_layer_att (:247-250)
    att_w[r]        [Cin_r, M]  'Conv2d_PrePose_Attn[r]' weights; NOTE the convs are CHAINED:
                                conv r consumes the OUTPUT of conv r-1 because `net` is
                                re-assigned at :263, so Cin_0 = C (or 768) and Cin_r = M after.
    td_w[r]         [C', K]     top-down convs 'Conv', 'Conv_1', ... (:298-303); C' = C (+J)
    Returns (logits [N,K], end_points) with the reference's end-point names (:287,:309,:352).
    """
    end_points: Dict[str, torch.Tensor] = {}
    net = last_conv if flags.single_layer_att else pose_pre_logits           # :247-250
    all_att = []
    for r in range(flags.rank):                                             # :258-270
        net = conv1x1(net, att_w[r], att_b[r])
        all_att.append(net)
    att = torch.stack(all_att, dim=-1) if len(all_att) > 1 else all_att[0]   # :271-274
    if flags.softmax_att:                                                   # :276-284
        att = spatial_softmax_nhwc(att)
    if flags.relu_att:                                                      # :285-286
        att = torch.relu(att)
    end_points['PosePrelogitsBasedAttention'] = att                         # :287

    feats = last_conv
    if flags.with_pose_feat:                                                # :289-295
        pl = pose_logits
        if flags.with_pose_feat_2layer:
            # :291-294 passes neither activation_fn nor normalizer_fn, so the resnet arg-scope
            # (resnet_utils.py:236-245) supplies relu AND slim.batch_norm (eps 1e-5, scale=True; a conv
            # with a normalizer has no bias).  This batch-norm sits outside resnet_v1()'s own
            # arg_scope([slim.batch_norm], is_training=...) (resnet_v1.py:191-194), so it runs with
            # slim's default is_training=True: batch statistics (tf.nn.moments, biased variance) in
            # training and in evaluation alike.  pose_feat_bn = (gamma, beta).
            # Under the other backbones' arg-scopes the same un-annotated conv is a different layer:
            #   vgg_arg_scope (vgg.py:49-62): activation relu, biases, no normalizer  -> relu(y + b)
            #   inception_v2_tsn_arg_scope (inception_v2_tsn.py:320-329): activation_fn=None,
            #   normalizer_fn=None                                                      -> y + b
            # (executed from the reference's own arg-scope functions in tests/golden/make_head_reference.py)
            #   inception_arg_scope (inception_utils.py:48-71, inception_v3): batch-norm with eps 1e-3 and NO gamma
            #   (scale defaults to False) + relu
            if arg_scope in ('resnet', 'inception_v3'):
                y = conv1x1(pl, pose_feat_w, None)
                mean = y.mean(dim=(0, 1, 2))
                var = y.var(dim=(0, 1, 2), unbiased=False)
                gamma, beta = pose_feat_bn
                y = (y - mean) / torch.sqrt(var + (1e-5 if arg_scope == 'resnet' else 1e-3))
                pl = torch.relu((y * gamma if gamma is not None else y) + beta)
            elif arg_scope == 'vgg':
                pl = torch.relu(conv1x1(pl, pose_feat_w, pose_feat_b))
            elif arg_scope == 'inception_v2_tsn':
                pl = conv1x1(pl, pose_feat_w, pose_feat_b)
            else:
                raise ValueError('arg_scope %r' % arg_scope)
        feats = torch.cat([feats, pl], dim=-1)
    feats = dropout(feats, keep_prob, dropout_mask, is_training)            # :296

    all_td = [conv1x1(feats, td_w[r], td_b[r]) for r in range(flags.rank)]  # :298-304
    td = torch.stack(all_td, dim=-1) if len(all_td) > 1 else all_td[0]      # :305-308
    end_points['TopDownAttention'] = td                                     # :309

    logits = (att * td).mean(dim=(1, 2), keepdim=True)                      # :322-325
    if logits.dim() == 5:                                                   # :326-328
        logits = logits.sum(dim=-1)
    logits = logits.squeeze(2).squeeze(1)                                   # :350-351
    end_points['Logits'] = logits                                           # :352
    return logits, end_points


def baseline_avgpool_logits(last_conv: torch.Tensor, w: torch.Tensor, b: torch.Tensor,
                            is_training: bool = False, keep_prob: float = 1.0,
                            dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cfg 001 (no attention): slim resnet_v1 head = global average pool + 1x1 conv 'logits'
    (models/slim/nets/resnet_v1.py:206-217; dropout patched in before the logits conv)."""
    net = last_conv.mean(dim=(1, 2), keepdim=True)
    net = dropout(net, keep_prob, dropout_mask, is_training)
    return conv1x1(net, w, b).squeeze(2).squeeze(1)


# --------------------------------------------------------------------------------------
# Frame pooling / temporal attention -- nets_factory.py:354-374
# --------------------------------------------------------------------------------------
def frame_pooling(logits: torch.Tensor, frames_per_video: int,
                  temporal_att_w: Optional[torch.Tensor] = None,
                  temporal_att_b: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """[B*F,K] -> [B,K].  tf.split into B groups of F consecutive rows (:359-361), optional
    `logits * conv1x1(logits;[K,1], bias init 1/F)` (:362-372), mean over frames (:374)."""
    ep: Dict[str, torch.Tensor] = {'logits_beforePool': logits}
    bf, k = logits.shape
    x = logits.reshape(bf // frames_per_video, frames_per_video, k)
    if temporal_att_w is not None:
        att = torch.matmul(x, temporal_att_w) + temporal_att_b              # [B,F,1]
        x = x * att
        ep['TemporalAttention'] = att.unsqueeze(-2)                         # [B,F,1,1] like :363-373
    return x.mean(dim=1), ep


# --------------------------------------------------------------------------------------
# TF1 legacy bilinear resize (align_corners=False, no half-pixel)  -- SURVEY Appendix B
# --------------------------------------------------------------------------------------
def tf1_resize_bilinear(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """tf.image.resize_images(..., BILINEAR) as of TF 1.1: src = dst * (in/out) in float32,
    lo = floor(src), hi = min(lo+1, in-1) (== min(ceil(src), in-1) whenever the lerp weight is non-zero), lerp.
    img: [N,H,W,C].
    Call sites: src/loss.py:21, src/preprocess_pipeline.py:204-207."""
    n, h, w, c = img.shape
    if (h, w) == (out_h, out_w):
        return img
    # the TF kernel (resize_bilinear_op.cc / image_resizer_state.h) computes scale and source coordinate in
    # FLOAT32: scale = in / float(out); src = float(dst) * scale -- restated with float32 tensors
    sy = torch.tensor(float(h), dtype=torch.float32) / torch.tensor(float(out_h), dtype=torch.float32)
    sx = torch.tensor(float(w), dtype=torch.float32) / torch.tensor(float(out_w), dtype=torch.float32)
    ys = torch.arange(out_h, dtype=torch.float32) * sy
    xs = torch.arange(out_w, dtype=torch.float32) * sx
    y0 = ys.floor().long(); y1 = torch.clamp(y0 + 1, max=h - 1); fy = (ys - ys.floor()).to(img.dtype)
    x0 = xs.floor().long(); x1 = torch.clamp(x0 + 1, max=w - 1); fx = (xs - xs.floor()).to(img.dtype)
    top = img[:, y0][:, :, x0] * (1 - fx)[None, None, :, None] + img[:, y0][:, :, x1] * fx[None, None, :, None]
    bot = img[:, y1][:, :, x0] * (1 - fx)[None, None, :, None] + img[:, y1][:, :, x1] * fx[None, None, :, None]
    return top * (1 - fy)[None, :, None, None] + bot * fy[None, :, None, None]


# --------------------------------------------------------------------------------------
# Losses -- src/loss.py
# --------------------------------------------------------------------------------------
def pose_l2_loss(logits_pose: torch.Tensor, labels_pose: torch.Tensor,
                 labels_pose_valid: torch.Tensor, pose_loss_wt: float = 1.0) -> torch.Tensor:
    """src/loss.py:11-70, `loss_type_pose == 'l2'`, LOSS_FN_POSE_SAMPLED off.

    Per keypoint channel j (:35; the loop header swaps the names lbl/lgt, harmless for L2):
        loss_val[n] = 0.5 * sum_hw (a-b)^2 / reduce_sum(ones([N,H,W]))        (:53-55)
    i.e. the divisor is N*H*W (it includes the batch!), then
        L_j = reduce_mean_n( where(valid[n,j], loss_val[n], 0) )              (:58-62)
    total = sum_j L_j * pose_loss_wt                                          (:69-70)
    => effective scale 0.5 / (N^2 * H * W).  Reproduced literally.
    labels are resized with the TF1 legacy bilinear rule if shapes differ (:14-22).
    """
    if labels_pose.shape != logits_pose.shape:
        labels_pose = tf1_resize_bilinear(labels_pose, logits_pose.shape[1], logits_pose.shape[2])
    n, h, w, j = logits_pose.shape
    mask_sum = float(n * h * w)                                             # reduce_sum(ones(shape(lgt)))
    total = logits_pose.new_zeros(())
    for ch in range(j):
        lbl = logits_pose[..., ch]          # names swapped exactly as :35 does
        lgt = labels_pose[..., ch]
        loss_val = 0.5 * ((lbl - lgt) ** 2).sum(dim=(1, 2)) / mask_sum
        v = labels_pose_valid[:, ch].to(torch.bool)
        L = torch.where(v, loss_val, torch.zeros_like(loss_val)).mean()
        total = total + L
    return total * pose_loss_wt


def action_softmax_xent(logits: torch.Tensor, labels: torch.Tensor, num_classes: int,
                        action_loss_wt: float = 1.0) -> torch.Tensor:
    """src/loss.py:74-80: tf.losses.softmax_cross_entropy(one_hot(labels,K), logits, weights=wt)
    = wt * sum_n( -log_softmax(logits)[n,label_n] ) / N   (SUM_BY_NONZERO_WEIGHTS, scalar weight)."""
    assert logits.shape[1] == num_classes
    lsm = torch.log_softmax(logits, dim=-1)
    per_ex = -lsm.gather(1, labels.view(-1, 1).long()).squeeze(1)
    if action_loss_wt == 0:
        return per_ex.sum() * 0.0
    return action_loss_wt * per_ex.sum() / logits.shape[0]


def action_l2(logits: torch.Tensor, labels: torch.Tensor, num_classes: int,
              action_loss_wt: float = 1.0) -> torch.Tensor:
    """src/loss.py:81-87: tf.losses.mean_squared_error(one_hot, logits, weights=wt): mean over
    all N*K elements times wt."""
    onehot = torch.nn.functional.one_hot(labels.long(), num_classes).to(logits.dtype)
    return action_loss_wt * ((logits - onehot) ** 2).mean()


def action_multi_label(logits: torch.Tensor, labels: torch.Tensor, pos_weight: float = 10.0) -> torch.Tensor:
    """src/loss.py:88-97: labels cast to float (multi-hot [N,K]);
    loss = reduce_mean(tf.nn.weighted_cross_entropy_with_logits(targets, logits, pos_weight=10)),
    registered with tf.losses.add_loss -- action_loss_wt is NOT applied by the reference.
    TF 1.x documents the op as  targets * -log(sigmoid(x)) * pos_weight + (1-targets) * -log(1-sigmoid(x))."""
    t = labels.to(logits.dtype)
    logsig = torch.nn.functional.logsigmoid
    per = t * -logsig(logits) * pos_weight + (1 - t) * -logsig(-logits)
    return per.mean()


def action_multi_label_2(logits: torch.Tensor, labels: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
    """src/loss.py:98-101: tf.losses.sigmoid_cross_entropy(multi_class_labels, logits), default
    weights=1.0: mean over all N*K elements of  max(x,0) - x*z + log(1 + exp(-|x|))."""
    t = labels.to(logits.dtype)
    per = torch.clamp(logits, min=0) - logits * t + torch.log1p(torch.exp(-logits.abs()))
    return weight * per.mean()


def pose_l2_sampled_loss(logits_pose: torch.Tensor, labels_pose: torch.Tensor,
                         labels_pose_valid: torch.Tensor, uniform: torch.Tensor,
                         pose_loss_wt: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """src/loss.py:29-70 with cfg.TRAIN.LOSS_FN_POSE_SAMPLED, line by line.  `uniform` [N,H,W,J] stands
    for the tf.random_uniform(tf.shape(lgt), 0, 1.0) draws of :43-45 (channel j of it for keypoint j).
    Returns (loss, PoseLossMask [N,H,W,J]).

    The loop header (:35) is `for v, lbl, lgt in zip(channels_valid, channels_logits, channels_labels)`:
    `lbl` holds the LOGITS channel and `lgt` the LABEL channel.  Restated with those names so that every
    line reads like the reference -- in particular `tf.greater(lbl, 0)` (:49) tests the logits."""
    if labels_pose.shape != logits_pose.shape:
        labels_pose = tf1_resize_bilinear(labels_pose, logits_pose.shape[1], logits_pose.shape[2])
    n, h, w, j = logits_pose.shape
    total = logits_pose.new_zeros(())
    masks = []
    for ch in range(j):
        v = labels_pose_valid[:, ch].to(torch.bool)
        lbl = logits_pose[..., ch]                                              # :35 (swapped names)
        lgt = labels_pose[..., ch]
        neg_areas = (lgt == 0)                                                  # :38
        pos_areas = (lgt > 0)                                                   # :39
        total_area = float(lgt.numel())                                         # :40  N*H*W
        pos_area_ratio = pos_areas.to(lgt.dtype).sum() / total_area             # :41
        neg_areas_selected = (uniform[..., ch].to(lgt.dtype) < pos_area_ratio).to(lgt.dtype) * \
            neg_areas.to(lgt.dtype)                                             # :43-46
        mask = ((neg_areas_selected + (lbl > 0).to(lgt.dtype)) > 0).to(lgt.dtype)   # :48-50
        lgt_m = lgt * mask                                                      # :51
        lbl_m = lbl * mask                                                      # :52
        loss_val = 0.5 * ((lbl_m - lgt_m) ** 2).mean(dim=(1, 2))                # :53
        masks.append(mask.unsqueeze(-1))                                        # :57
        L = torch.where(v, loss_val, torch.zeros_like(loss_val)).mean()         # :58-62
        total = total + L
    return total * pose_loss_wt, torch.cat(masks, dim=-1)                       # :68-70


def l2_regularizer(weights: Sequence[torch.Tensor], weight_decay: float) -> torch.Tensor:
    """slim.l2_regularizer(wd)(W) = wd * tf.nn.l2_loss(W) = wd * 0.5 * sum(W^2), conv weights
    only (models/slim/nets/resnet_utils.py:241); biases are not regularised."""
    tot = weights[0].new_zeros(())
    for w in weights:
        tot = tot + weight_decay * 0.5 * (w ** 2).sum()
    return tot


def gen_losses(labels_action, logits_action, loss_type_action, num_action_classes, action_loss_wt,
               labels_pose, logits_pose, loss_type_pose, labels_pose_valid, pose_loss_wt,
               end_points=None, cfg=None) -> List[torch.Tensor]:
    """Same 12-argument call surface as src/loss.py:4-8; returns the list of losses that the
    reference adds to tf.GraphKeys.LOSSES (pose first, then action)."""
    losses: List[torch.Tensor] = []
    if loss_type_pose and logits_pose is not None and logits_pose.shape[-1] > 0:
        if loss_type_pose != 'l2':
            raise ValueError('Invalid loss {}'.format(loss_type_pose))
        if cfg is not None and cfg.TRAIN.LOSS_FN_POSE_SAMPLED:                      # loss.py:36-52
            loss_p, mask = pose_l2_sampled_loss(logits_pose, labels_pose, labels_pose_valid,
                                                end_points['PoseLossUniform'], pose_loss_wt)
            end_points['PoseLossMask'] = mask
            losses.append(loss_p)
        else:
            losses.append(pose_l2_loss(logits_pose, labels_pose, labels_pose_valid, pose_loss_wt))
            if end_points is not None:                                              # loss.py:54,57,68
                end_points['PoseLossMask'] = torch.ones_like(logits_pose)
    if loss_type_action == 'softmax-xentropy':
        losses.append(action_softmax_xent(logits_action, labels_action, num_action_classes, action_loss_wt))
    elif loss_type_action == 'l2':
        losses.append(action_l2(logits_action, labels_action, num_action_classes, action_loss_wt))
    elif loss_type_action == 'multi-label':                                         # loss.py:88-97
        losses.append(action_multi_label(logits_action, labels_action))
    elif loss_type_action == 'multi-label-2':                                       # loss.py:98-101
        losses.append(action_multi_label_2(logits_action, labels_action))
    elif loss_type_action == '':
        pass
    else:
        raise ValueError('Unrecognized loss {}'.format(loss_type_action))
    return losses


# --------------------------------------------------------------------------------------
# Data-parallel deployment semantics -- models/slim/deployment/model_deploy.py
# --------------------------------------------------------------------------------------
def dp_clone_loss(clone_losses: Sequence[torch.Tensor], num_clones: int) -> torch.Tensor:
    """_gather_clone_loss (:200-238): add_n of the clone's losses, divided by num_clones."""
    tot = clone_losses[0]
    for l in clone_losses[1:]:
        tot = tot + l
    return tot / float(num_clones) if num_clones > 1 else tot


def dp_sum_clone_grads(clone_grads: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """_sum_clones_gradients (:421-451): per variable, add_n of the tower gradients."""
    out = []
    for per_var in zip(*clone_grads):
        g = per_var[0].clone()
        for other in per_var[1:]:
            g = g + other
        out.append(g)
    return out


# --------------------------------------------------------------------------------------
# Parameter bundles + an end-to-end "head" forward used by the tests and the CPU baseline
# --------------------------------------------------------------------------------------
@dataclass
class HeadParams:
    att_w: List[torch.Tensor]
    att_b: List[torch.Tensor]
    td_w: List[torch.Tensor]
    td_b: List[torch.Tensor]
    pose_w1: Optional[torch.Tensor] = None
    pose_b1: Optional[torch.Tensor] = None
    pose_w2: Optional[torch.Tensor] = None
    pose_b2: Optional[torch.Tensor] = None

    def leaves(self) -> List[torch.Tensor]:
        xs = list(self.att_w) + list(self.att_b) + list(self.td_w) + list(self.td_b)
        for t in (self.pose_w1, self.pose_b1, self.pose_w2, self.pose_b2):
            if t is not None:
                xs.append(t)
        return xs


def head_forward(x: torch.Tensor, p: HeadParams, flags: AttnFlags, *, is_training=False,
                 keep_prob=0.2, dropout_mask=None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """network_fn's head: PoseLogits (always built, :147-160) + attentional pooling."""
    ep: Dict[str, torch.Tensor] = {}
    pre = pl = None
    if p.pose_w1 is not None:
        pre, pl = pose_logits_head(x, p.pose_w1, p.pose_b1, p.pose_w2, p.pose_b2)
        ep['PoseLogits'] = pl
    logits, ep2 = attentional_pooling(x, pre, pl, p.att_w, p.att_b, p.td_w, p.td_b, flags,
                                      is_training=is_training, keep_prob=keep_prob,
                                      dropout_mask=dropout_mask)
    ep.update(ep2)
    return logits, ep
