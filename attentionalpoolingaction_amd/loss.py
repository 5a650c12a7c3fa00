"""Losses of the head, behind the reference's `gen_losses` call surface.

Reference: /root/reference/src/loss.py:4-105.  Every branch is a HIP kernel that returns value and
gradient from one launch (include/apa.h):
  * pose 'l2' with per-keypoint validity mask (loss.py:29-70)   -> apa_pose_l2_loss_fwd_bwd
  * the same with LOSS_FN_POSE_SAMPLED (loss.py:36-52)           -> apa_pose_sampled_loss_fwd_bwd
  * action 'softmax-xentropy' (loss.py:74-80)                    -> apa_softmax_xent_fwd_bwd
  * action 'l2' / 'multi-label' / 'multi-label-2' (loss.py:81-101) -> apa_action_loss_fwd_bwd
  * the label resize of loss.py:14-22                            -> apa_resize_bilinear_tf1
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .custom_ops import custom_ops_factory as cof


class SoftmaxXentFunction(torch.autograd.Function):
    """loss = wt * mean_n( -log_softmax(logits)[n, labels[n]] )  (tf.losses.softmax_cross_entropy
    with a scalar weight).  The gradient is produced by the same kernel launch."""

    @staticmethod
    def forward(ctx, logits, labels, wt):
        lossbuf, G, _, _ = cof.softmax_xent_fwd_bwd(logits.contiguous(), labels.contiguous(), wt=wt,
                                                    grad_scale=1.0, want_grad=True)
        ctx.save_for_backward(G)
        return lossbuf[0]

    @staticmethod
    def backward(ctx, dloss):
        (G,) = ctx.saved_tensors
        return G * dloss, None, None


class PoseL2Function(torch.autograd.Function):
    """loss.py:53-70 literally: sum_j mean_n( valid[n,j] ? 0.5*sum_hw (a-b)^2 / (N*H*W) : 0 ) * wt."""

    @staticmethod
    def forward(ctx, logits_pose, labels_pose, valid, wt):
        loss, dPl = cof.pose_l2_loss_fwd_bwd(logits_pose.contiguous(), labels_pose.contiguous(),
                                             valid, wt=wt, grad_scale=1.0, want_grad=True)
        ctx.save_for_backward(dPl)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dPl,) = ctx.saved_tensors
        return dPl * dloss, None, None, None


class ActionLossFunction(torch.autograd.Function):
    """'l2' / 'multi-label' / 'multi-label-2' (loss.py:81-101): value + gradient from one launch."""

    @staticmethod
    def forward(ctx, logits, labels, kind, wt):
        loss, G = cof.action_loss_fwd_bwd(kind, logits.contiguous().float(), labels.contiguous(), wt=wt)
        ctx.save_for_backward(G)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (G,) = ctx.saved_tensors
        return G * dloss, None, None, None


class PoseSampledFunction(torch.autograd.Function):
    """LOSS_FN_POSE_SAMPLED (loss.py:36-52); returns (loss, mask) -- the mask is the PoseLossMask end point."""

    @staticmethod
    def forward(ctx, logits_pose, labels_pose, valid, uniform, wt):
        loss, dPl, mask = cof.pose_sampled_loss_fwd_bwd(logits_pose.contiguous(), labels_pose.contiguous(),
                                                        valid, uniform.contiguous(), wt=wt)
        ctx.save_for_backward(dPl)
        ctx.mark_non_differentiable(mask)
        return loss[0], mask

    @staticmethod
    def backward(ctx, dloss, _dmask):
        (dPl,) = ctx.saved_tensors
        return dPl * dloss, None, None, None, None


def tf1_resize_bilinear(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """tf.image.resize_images as of TF 1.1 (legacy bilinear, no half-pixel offset); used by
    loss.py:21 when the label map and the pose logits disagree in size -- apa_resize_bilinear_tf1
    (labels carry no gradient)."""
    if (img.shape[1], img.shape[2]) == (out_h, out_w):
        return img
    return cof.resize_bilinear_tf1(img.contiguous().float(), out_h, out_w)


def gen_losses(labels_action, logits_action, loss_type_action, num_action_classes, action_loss_wt,
               labels_pose, logits_pose, loss_type_pose, labels_pose_valid, pose_loss_wt,
               end_points: Optional[Dict[str, torch.Tensor]] = None, cfg=None) -> List[torch.Tensor]:
    """Same 12 positional arguments as src/loss.py:4-8.  The reference registers the losses in
    tf.GraphKeys.LOSSES; here they are returned (pose first, then action) and the caller sums
    them, exactly what model_deploy._gather_clone_loss does with the collection."""
    losses: List[torch.Tensor] = []
    if loss_type_pose and logits_pose is not None and logits_pose.shape[-1] > 0:
        if loss_type_pose != 'l2':
            raise ValueError('Invalid loss {}'.format(loss_type_pose))
        if labels_pose.shape != logits_pose.shape:
            labels_pose = tf1_resize_bilinear(labels_pose, logits_pose.shape[1], logits_pose.shape[2])
        if cfg is not None and cfg.TRAIN.LOSS_FN_POSE_SAMPLED:
            # tf.random_uniform(tf.shape(lgt), 0, 1) per keypoint channel (loss.py:43-45); a caller that
            # needs the draws (parity tests) passes them as end_points['PoseLossUniform']
            u = None if end_points is None else end_points.get('PoseLossUniform')
            if u is None:
                u = torch.rand(logits_pose.shape, device=logits_pose.device, dtype=torch.float32)
            loss_p, mask = PoseSampledFunction.apply(logits_pose.float(), labels_pose.float(),
                                                     labels_pose_valid, u, float(pose_loss_wt))
            losses.append(loss_p)
            if end_points is not None:
                end_points['PoseLossMask'] = mask.view(logits_pose.shape)            # loss.py:64,68
        else:
            losses.append(PoseL2Function.apply(logits_pose.float(), labels_pose.float(),
                                               labels_pose_valid, float(pose_loss_wt)))
            if end_points is not None:
                end_points['PoseLossMask'] = torch.ones_like(logits_pose)     # loss.py:54,57,68
    if loss_type_action == 'softmax-xentropy':
        assert logits_action.shape[1] == num_action_classes
        losses.append(SoftmaxXentFunction.apply(logits_action, labels_action.long(),
                                                float(action_loss_wt)))
    elif loss_type_action == '':
        pass
    elif loss_type_action == 'l2':                                            # loss.py:81-87
        assert logits_action.shape[1] == num_action_classes
        losses.append(ActionLossFunction.apply(logits_action, labels_action.long(), 'l2',
                                               float(action_loss_wt)))
    elif loss_type_action in ('multi-label', 'multi-label-2'):                # loss.py:88-101
        # multi-hot float labels; 'multi-label' ignores action_loss_wt (tf.losses.add_loss on the bare
        # mean), 'multi-label-2' is tf.losses.sigmoid_cross_entropy with its default weight 1.0
        losses.append(ActionLossFunction.apply(logits_action, labels_action.float(), loss_type_action, 1.0))
    else:
        raise ValueError('Unrecognized loss {}'.format(loss_type_action))
    return losses


def l2_regularization(weights, weight_decay: float) -> torch.Tensor:
    """slim.l2_regularizer(wd): wd * 0.5 * sum(W^2) over conv weights (resnet_utils.py:241)."""
    tot = None
    for w in weights:
        t = 0.5 * weight_decay * (w.float() ** 2).sum()
        tot = t if tot is None else tot + t
    return tot
