"""Losses of the head, behind the reference's `gen_losses` call surface.

Reference: /root/reference/src/loss.py:4-105.  The two losses the shipped configs use are fused
HIP kernels (value + gradient in one launch, see include/apa.h):
  * pose 'l2' with per-keypoint validity mask (loss.py:29-70)   -> apa_pose_l2_loss_fwd_bwd
  * action 'softmax-xentropy' (loss.py:74-80)                    -> apa_softmax_xent_fwd_bwd
Anything else the reference accepts ('l2' action loss, multi-label variants, the sampled pose
loss) raises NotImplementedError -- no shipped experiment selects them.
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


def tf1_resize_bilinear(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """tf.image.resize_images as of TF 1.1 (legacy bilinear, no half-pixel offset); used by
    loss.py:21 when the label map and the pose logits disagree in size.  Device-agnostic torch ops
    on a tiny label tensor (plumbing, not the hot path)."""
    n, h, w, c = img.shape
    if (h, w) == (out_h, out_w):
        return img
    ys = torch.arange(out_h, dtype=torch.float64, device=img.device) * (h / out_h)
    xs = torch.arange(out_w, dtype=torch.float64, device=img.device) * (w / out_w)
    y0 = ys.floor().long(); y1 = torch.clamp(y0 + 1, max=h - 1); fy = (ys - y0).to(img.dtype)
    x0 = xs.floor().long(); x1 = torch.clamp(x0 + 1, max=w - 1); fx = (xs - x0).to(img.dtype)
    fxv = fx[None, None, :, None]
    top = img[:, y0][:, :, x0] * (1 - fxv) + img[:, y0][:, :, x1] * fxv
    bot = img[:, y1][:, :, x0] * (1 - fxv) + img[:, y1][:, :, x1] * fxv
    fyv = fy[None, :, None, None]
    return top * (1 - fyv) + bot * fyv


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
        if cfg is not None and cfg.TRAIN.LOSS_FN_POSE_SAMPLED:
            raise NotImplementedError('LOSS_FN_POSE_SAMPLED (loss.py:36-52) is off in every shipped '
                                      'config and not built')
        if labels_pose.shape != logits_pose.shape:
            labels_pose = tf1_resize_bilinear(labels_pose, logits_pose.shape[1], logits_pose.shape[2])
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
    elif loss_type_action in ('l2', 'multi-label', 'multi-label-2'):
        raise NotImplementedError("action loss '{}' (loss.py:81-101) is not used by any shipped "
                                  'config and not built'.format(loss_type_action))
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
