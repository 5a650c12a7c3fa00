"""The attentional-pooling head behind the reference's model-factory call surface.

Reference: /root/reference/models/slim/nets/nets_factory.py:94-380 --
`get_network_fn(name, num_classes, num_pose_keypoints, cfg, weight_decay, is_training)` returns
`network_fn(images) -> (logits [N,K], end_points)`.  Here the ResNet-101 backbone is an optional
callable (PyTorch-ROCm, out of this hot path); without one `network_fn` takes the conv5 feature
map `last_conv` [N,H,W,C] (NHWC, float32 or bfloat16, resident in HBM) directly.  The head itself
runs in hand-written HIP (libapa_hip.so) through torch.autograd.Function wrappers -- torch only
carries device memory, the stream and the autograd graph.

End-point names are the reference's: 'PosePrelogitsBasedAttention' (:287), 'Logits' (:352),
'logits_beforePool' (:357).  Parameter names map 1:1 onto the TF variable names.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .config import dropout_keep_prob
from .custom_ops import custom_ops_factory as cof

last_conv_map = {  # nets_factory.py:63-67 ; channel counts of the taps
    'inception_v3': ('Mixed_7c', 2048),
    'inception_v2_tsn': ('InceptionV2_TSN/inception_5b', 1024),
    'resnet_v1_101': ('resnet_v1_101/block4', 2048),
    'vgg_16': ('vgg_16/conv5', 512),
}


class AttentionalPoolingFunction(torch.autograd.Function):
    """logits, att = f(X, Xatt, Wa, ba, Wt, bt); nets_factory.py:247-328 in two HIP calls.

    `Xatt is None` means the attention map is computed from X itself (cfg 002,
    ..._SINGLE_LAYER_ATT).  `att` (the 'PosePrelogitsBasedAttention' end point) is returned for
    inspection and is not differentiable on its own, exactly as nothing in the reference puts a
    loss on it.
    """

    @staticmethod
    def forward(ctx, X, Xatt, Wa, ba, Wt, bt, flags, keep_prob, seed, offset):
        Xc = X.contiguous()
        fused = Xatt is None
        Xa = Xc if fused else Xatt.contiguous()
        logits, att, zsave, abar, _, ws = cof.attn_pool_fwd(
            Xc, Xa, Wa.contiguous(), ba.contiguous(), Wt.contiguous(), bt.contiguous(),
            flags=flags, keep_prob=keep_prob, seed=seed, offset=offset)
        ctx.save_for_backward(Xc, Xa if not fused else Xc, Wa, ba, Wt, bt, att, zsave, abar)
        ctx.fused = fused
        ctx.cfg = (flags, keep_prob, seed, offset)
        ctx.ws = ws
        ctx.xshape = X.shape
        ctx.xatt_shape = None if fused else Xatt.shape
        ctx.mark_non_differentiable(att)
        return logits, att

    @staticmethod
    def backward(ctx, dlogits, _datt):
        Xc, Xa, Wa, ba, Wt, bt, att, zsave, abar = ctx.saved_tensors
        flags, keep_prob, seed, offset = ctx.cfg
        if ctx.fused:
            Xa = Xc
        dX, dXatt, dWa, dba, dWt, dbt = cof.attn_pool_bwd(
            Xc, Xa, Wa.contiguous(), ba.contiguous(), Wt.contiguous(), bt.contiguous(), att, zsave,
            abar, dlogits.contiguous().float(), flags=flags, keep_prob=keep_prob, seed=seed,
            offset=offset, workspace=ctx.ws)
        dX = dX.view(ctx.xshape)
        if dXatt is not None:
            dXatt = dXatt.view(ctx.xatt_shape)
        return dX, dXatt, dWa, dba, dWt, dbt, None, None, None, None


def attentional_pooling(X, Xatt, Wa, ba, Wt, bt, *, softmax_att=False, relu_att=False,
                        is_training=False, keep_prob=0.2, seed=0, offset=0):
    flags = cof.attn_flags(softmax_att, relu_att, is_training)
    return AttentionalPoolingFunction.apply(X, Xatt, Wa, ba, Wt, bt, flags,
                                            keep_prob if is_training else 1.0, seed, offset)


class AttentionalPoolingHead(nn.Module):
    """The `USE_POSE_PRELOGITS_BASED_ATTENTION` head (nets_factory.py:242-352) as a module.

    Parameters (TF variable name -> attribute):
      PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/{weights,biases} -> att_weights [Cin,M], att_biases [M]
      PosePrelogitsBasedAttention/Conv/{weights,biases}                -> td_weights [C,K],   td_biases [K]
    Initialisation follows the reference: weights ~ N(0, 0.001), biases zero (:141,265-266,301-302).
    """

    TF_NAMES = {
        'att_weights': 'PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/weights',
        'att_biases': 'PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/biases',
        'td_weights': 'PosePrelogitsBasedAttention/Conv/weights',
        'td_biases': 'PosePrelogitsBasedAttention/Conv/biases',
    }

    def __init__(self, num_classes: int, cfg, in_channels: int = 2048, att_in_channels: int = None,
                 is_training: bool = False, seed: int = 42):
        super().__init__()
        net = cfg.NET
        if not net.USE_POSE_PRELOGITS_BASED_ATTENTION:
            raise ValueError('AttentionalPoolingHead needs cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION')
        if net.USE_POSE_PRELOGITS_BASED_ATTENTION_RANK != 1:
            raise NotImplementedError('rank > 1 (chained attention convs, nets_factory.py:258-274) '
                                      'is not built; every shipped config uses rank 1')
        if net.USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT:
            raise NotImplementedError('..._WITH_POSE_FEAT (nets_factory.py:289-295) is off in all '
                                      'shipped configs and not built')
        self.num_classes = num_classes
        self.single_layer = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT)
        self.softmax_att = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT)
        self.relu_att = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_RELU_ATT)
        self.per_class = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS)
        self.keep_prob = dropout_keep_prob(cfg)
        self.is_training = is_training
        self.seed = seed
        self._step = 0
        n_maps = num_classes if self.per_class else 1
        cin = in_channels if self.single_layer else (att_in_channels or 768)
        self.att_weights = nn.Parameter(torch.randn(cin, n_maps) * 0.001)
        self.att_biases = nn.Parameter(torch.zeros(n_maps))
        self.td_weights = nn.Parameter(torch.randn(in_channels, num_classes) * 0.001)
        self.td_biases = nn.Parameter(torch.zeros(num_classes))

    def regularized_weights(self):
        """conv weights carry slim.l2_regularizer from the resnet arg-scope (resnet_utils.py:241);
        biases do not."""
        return [self.att_weights, self.td_weights]

    def forward(self, last_conv: torch.Tensor, pose_pre_logits: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        end_points: Dict[str, torch.Tensor] = {}
        if not self.single_layer and pose_pre_logits is None:
            raise ValueError('cfg selects attention from pose_pre_logits (cfg 003) but none given')
        xatt = None if self.single_layer else pose_pre_logits
        offset = self._step
        if self.is_training:
            self._step += 1            # a fresh dropout mask per step
        logits, att = attentional_pooling(
            last_conv, xatt, self.att_weights, self.att_biases, self.td_weights, self.td_biases,
            softmax_att=self.softmax_att, relu_att=self.relu_att, is_training=self.is_training,
            keep_prob=self.keep_prob, seed=self.seed, offset=offset)
        n, h, w = last_conv.shape[0], last_conv.shape[1], last_conv.shape[2]
        end_points['PosePrelogitsBasedAttention'] = att.view(n, h, w, -1)
        end_points['Logits'] = logits
        return logits, end_points


def frame_pooling(logits: torch.Tensor, frames_per_video: int, end_points: Dict[str, torch.Tensor]
                  ) -> torch.Tensor:
    """nets_factory.py:354-374 without temporal attention: [B*F,K] -> mean over the F frames."""
    end_points['logits_beforePool'] = logits
    bf, k = logits.shape
    return logits.view(bf // frames_per_video, frames_per_video, k).mean(dim=1)


def get_network_fn(name: str, num_classes: int, num_pose_keypoints: int, cfg,
                   weight_decay: float = 0.0, is_training: bool = False,
                   backbone: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                   device='cuda'):
    """Same signature as nets_factory.py:94-95 (+ optional backbone/device).

    Returns `network_fn(images) -> (logits, end_points)`; `network_fn.head` exposes the module
    (its parameters) and `network_fn.weight_decay` the coefficient for the L2 term.
    With `backbone=None`, `images` IS the conv5 map [N,H,W,C] (or [B,F,H,W,C] for video input,
    nets_factory.py:121-125).
    """
    if name not in last_conv_map:
        raise ValueError('Name of network unknown %s' % name)
    channels = last_conv_map[name][1]
    head = AttentionalPoolingHead(num_classes, cfg, in_channels=channels,
                                  is_training=is_training, seed=cfg.RNG_SEED).to(device)

    def network_fn(images: torch.Tensor, pose_pre_logits: Optional[torch.Tensor] = None):
        frames_per_video = 1
        if images.dim() == 5:                                   # :121-125
            frames_per_video = images.shape[1]
            images = images.reshape(-1, *images.shape[2:])
        last_conv = backbone(images) if backbone is not None else images
        logits, end_points = head(last_conv, pose_pre_logits)
        if frames_per_video > 1:                                # :354-374
            logits = frame_pooling(logits, frames_per_video, end_points)
        return logits, end_points

    network_fn.head = head
    network_fn.weight_decay = weight_decay
    network_fn.num_pose_keypoints = num_pose_keypoints
    return network_fn
