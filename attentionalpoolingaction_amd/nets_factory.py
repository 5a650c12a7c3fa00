"""The attentional-pooling head behind the reference's model-factory call surface.

Reference: /root/reference/models/slim/nets/nets_factory.py:94-380 --
`get_network_fn(name, num_classes, num_pose_keypoints, cfg, weight_decay, is_training)` returns
`network_fn(images) -> (logits [N,K], end_points)`.  Here the ResNet-101 backbone is an optional
callable (PyTorch-ROCm, out of this hot path); without one `network_fn` takes the conv5 feature
map `last_conv` [N,H,W,C] (NHWC, float32 or bfloat16, resident in HBM) directly.  The head itself
runs in hand-written HIP (libapa_hip.so) through torch.autograd.Function wrappers -- torch only
carries device memory, the stream and the autograd graph.

End-point names are the reference's: 'PoseLogits' (:160), 'PosePrelogitsBasedAttention' (:287),
'TopDownAttention' (:309, on request), 'Logits' (:352), 'logits_beforePool' (:357).
Parameter names map 1:1 onto the TF variable names.
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


class EndPoints(dict):
    """The end-point dictionary network_fn returns.  A TF graph evaluates an end point only when it is fetched;
    `lazy(name, thunk)` gives an entry the same behaviour here: it is computed (and cached) on first access.
    Used for end_points['PoseLogits'], which the reference builds for every configuration (nets_factory.py:
    147-160) and reads unconditionally (src/train.py:400) but TF prunes whenever no loss or attention input
    consumes it (cfg 002): the pose head's 617 MFLOP/image only run if somebody actually looks."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = {}

    def lazy(self, name, thunk):
        self._lazy[name] = thunk

    def __missing__(self, name):
        if name in self._lazy:
            value = self._lazy.pop(name)()
            self[name] = value
            return value
        raise KeyError(name)

    def __contains__(self, name):
        return dict.__contains__(self, name) or name in self._lazy

    def get(self, name, default=None):
        return self[name] if name in self else default

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._lazy if not dict.__contains__(self, k)]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    # every view agrees with keys(): a lazy entry is evaluated by whoever walks over the values
    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def copy(self):
        c = EndPoints(dict.items(self))
        c._lazy = dict(self._lazy)
        return c

    def pop(self, name, *default):
        if dict.__contains__(self, name):
            return dict.pop(self, name)
        if name in self._lazy:
            return self._lazy.pop(name)()
        if default:
            return default[0]
        raise KeyError(name)


ARG_SCOPE_OF = {  # arg_scopes_map (nets_factory.py:69-91): which slim defaults surround the head
    'inception_v3': 'inception_v3', 'inception_v2_tsn': 'inception_v2_tsn', 'resnet_v1_101': 'resnet',
    'vgg_16': 'vgg',
}


class AttentionalPoolingFunction(torch.autograd.Function):
    """logits, att[, topdown] = f(X, Xatt, Wa, ba, Wt, bt); nets_factory.py:247-328 in two HIP calls.

    `Xatt is None` means the attention map is computed from X itself (cfg 002,
    ..._SINGLE_LAYER_ATT).  Wa [Cin,1] selects the factorised class-agnostic path, Wa [Cin,K] the
    dense per-class path.  `att` / `topdown` are end points returned for inspection; nothing in
    the reference puts a loss on them, so they are marked non-differentiable.
    """

    @staticmethod
    def forward(ctx, X, Xatt, Wa, ba, Wt, bt, flags, keep_prob, seed, offset, want_topdown, hooks=None):
        Xc = X.contiguous()
        fused = Xatt is None
        Xa = Xc if fused else Xatt.contiguous()
        logits, att, zsave, abar, topdown, ws = cof.attn_pool_fwd(
            Xc, Xa, Wa.contiguous(), ba.contiguous(), Wt.contiguous(), bt.contiguous(),
            flags=flags, keep_prob=keep_prob, seed=seed, offset=offset, want_topdown=want_topdown, hooks=hooks)
        ctx.hooks = hooks
        saved = [Xc, Xa, Wa, ba, Wt, bt, att, zsave]
        ctx.has_abar = abar is not None
        if abar is not None:
            saved.append(abar)
        ctx.save_for_backward(*saved)
        ctx.fused = fused
        ctx.cfg = (flags, keep_prob, seed, offset)
        ctx.ws = ws
        ctx.xshape = X.shape
        ctx.xatt_shape = None if fused else Xatt.shape
        ctx.mark_non_differentiable(att)
        if topdown is not None:
            ctx.mark_non_differentiable(topdown)
            return logits, att, topdown
        return logits, att, None

    @staticmethod
    def backward(ctx, dlogits, _datt, _dtd):
        saved = ctx.saved_tensors
        Xc, Xa, Wa, ba, Wt, bt, att, zsave = saved[:8]
        abar = saved[8] if ctx.has_abar else None
        flags, keep_prob, seed, offset = ctx.cfg
        if ctx.fused:
            Xa = Xc
        dX, dXatt, dWa, dba, dWt, dbt = cof.attn_pool_bwd(
            Xc, Xa, Wa.contiguous(), ba.contiguous(), Wt.contiguous(), bt.contiguous(), att, zsave,
            abar, dlogits.contiguous().float(), flags=flags, keep_prob=keep_prob, seed=seed,
            offset=offset, workspace=ctx.ws, hooks=ctx.hooks)
        dX = dX.view(ctx.xshape)
        if dXatt is not None:
            dXatt = dXatt.view(ctx.xatt_shape)
        return dX, dXatt, dWa, dba, dWt, dbt, None, None, None, None, None, None


class AttentionalPoolingCatFunction(torch.autograd.Function):
    """logits, att = f(X, Xatt, Xext, Wa, ba, Wt, bt): ..._WITH_POSE_FEAT (nets_factory.py:289-296) --
    attentional pooling over concat(X, Xext) with Wt [C+J, K], the concatenation never formed
    (apa_attn_pool_fwd_cat / apa_attn_pool_bwd_cat); `Xatt is None` = attention from X itself."""

    @staticmethod
    def forward(ctx, X, Xatt, Xext, Wa, ba, Wt, bt, flags, keep_prob, seed, offset, hooks=None):
        Xc = X.contiguous()
        fused = Xatt is None
        Xa = Xc if fused else Xatt.contiguous()
        Xe = Xext.contiguous().float()
        logits, att, zsave, abar, zext, ws = cof.attn_pool_fwd_cat(
            Xc, Xa, Xe, Wa.contiguous(), ba.contiguous(), Wt.contiguous(), bt.contiguous(), flags=flags,
            keep_prob=keep_prob, seed=seed, offset=offset, hooks=hooks)
        ctx.hooks = hooks
        ctx.save_for_backward(Xc, Xa, Xe, Wa, ba, Wt, bt, att, zsave, abar, zext)
        ctx.fused = fused
        ctx.cfg = (flags, keep_prob, seed, offset)
        ctx.ws = ws
        ctx.shapes = (X.shape, None if fused else Xatt.shape, Xext.shape)
        ctx.mark_non_differentiable(att)
        return logits, att

    @staticmethod
    def backward(ctx, dlogits, _datt):
        Xc, Xa, Xe, Wa, ba, Wt, bt, att, zsave, abar, zext = ctx.saved_tensors
        flags, keep_prob, seed, offset = ctx.cfg
        if ctx.fused:
            Xa = Xc
        dX, dXatt, dXext, dWa, dba, dWt, dbt = cof.attn_pool_bwd_cat(
            Xc, Xa, Xe, zext, Wa.contiguous(), ba.contiguous(), Wt.contiguous(), bt.contiguous(), att, zsave,
            abar, dlogits.contiguous().float(), flags=flags, keep_prob=keep_prob, seed=seed, offset=offset,
            workspace=ctx.ws, hooks=ctx.hooks)
        xs, xas, xes = ctx.shapes
        return (dX.view(xs), None if dXatt is None else dXatt.view(xas), dXext.view(xes), dWa, dba, dWt, dbt,
                None, None, None, None, None)


class PoseHeadFunction(torch.autograd.Function):
    """Ppre, Pl = f(X, W1, b1, W2, b2): the PoseLogits head (nets_factory.py:147-160) on MFMA."""

    @staticmethod
    def forward(ctx, X, W1, b1, W2, b2):
        Xc = X.contiguous()
        Ppre, Pl, ws = cof.pose_head_fwd(Xc, W1.contiguous(), b1.contiguous(), W2.contiguous(),
                                         b2.contiguous())
        ctx.save_for_backward(Xc, W1, W2, Ppre)
        ctx.ws = ws
        ctx.set_materialize_grads(False)   # an unused output arrives as None, not as zeros
        return Ppre, Pl

    @staticmethod
    def backward(ctx, dPpre, dPl):
        if dPpre is None and dPl is None:
            return None, None, None, None, None
        Xc, W1, W2, Ppre = ctx.saved_tensors
        dPl_c = None if dPl is None else dPl.contiguous().float()
        dPpre_c = None if dPpre is None else dPpre.contiguous().to(Xc.dtype)
        dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xc, W1.contiguous(), W2.contiguous(), Ppre, dPl_c,
                                                   dPpre_c, workspace=ctx.ws, ws_from_fwd=True)
        return dX, dW1, db1, dW2, db2


class PoseAttentionFunction(torch.autograd.Function):
    """cfg 003 in one autograd node: PoseLogits head -> attention map from its pre-logits -> pooling.
    logits, att, Pl[, topdown] = f(X, W1, b1, W2, b2, Wa, ba, Wt, bt).

    Keeping the two ops in one node lets the backward pass hand the attention-branch gradient to the
    pose head in rank-1 form (dXatt = dZ (x) Wa: APA_FLAG_DXATT_RANK1 + apa_pose_head_bwd_rank1ext --
    the [N,P,768] tensor is neither written nor read) and lets the pose head ADD its dX into the
    pooling op's buffer instead of autograd summing two [N,P,2048] tensors."""

    @staticmethod
    def forward(ctx, X, W1, b1, W2, b2, Wa, ba, Wt, bt, flags, keep_prob, seed, offset, want_topdown, hooks=None):
        Xc = X.contiguous()
        W1c, W2c, Wac, Wtc = W1.contiguous(), W2.contiguous(), Wa.contiguous(), Wt.contiguous()
        Ppre, Pl, pws = cof.pose_head_fwd(Xc, W1c, b1.contiguous(), W2c, b2.contiguous())
        logits, att, zsave, abar, topdown, aws = cof.attn_pool_fwd(
            Xc, Ppre, Wac, ba.contiguous(), Wtc, bt.contiguous(), flags=flags, keep_prob=keep_prob,
            seed=seed, offset=offset, want_topdown=want_topdown, hooks=hooks)
        ctx.hooks = hooks
        ctx.save_for_backward(Xc, W1c, W2c, Ppre, Wac, ba, Wtc, bt, att, zsave, abar)
        ctx.cfg = (flags, keep_prob, seed, offset)
        ctx.ws = (pws, aws)
        ctx.xshape = X.shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(att)
        if topdown is not None:
            ctx.mark_non_differentiable(topdown)
        return logits, att, Pl, topdown

    @staticmethod
    def backward(ctx, dlogits, _datt, dPl, _dtd):
        Xc, W1, W2, Ppre, Wa, ba, Wt, bt, att, zsave, abar = ctx.saved_tensors
        flags, keep_prob, seed, offset = ctx.cfg
        pws, aws = ctx.ws
        dPl_c = None if dPl is None else dPl.contiguous().float()
        if dlogits is None and dPl_c is None:
            return (None,) * 15
        dX = dZ = dWa = dba = dWt = dbt = None
        if dlogits is not None:
            epv = 4 if Xc.dtype == torch.float32 else 8
            rank1 = Wa.shape[0] % epv == 0 and Wa.shape[0] // epv <= 256   # the register-resident GEMV
            dX, dZ, dWa, dba, dWt, dbt = cof.attn_pool_bwd(
                Xc, Ppre, Wa, ba.contiguous(), Wt, bt.contiguous(), att, zsave, abar,
                dlogits.contiguous().float(), flags=flags, keep_prob=keep_prob, seed=seed, offset=offset,
                workspace=aws, dxatt_rank1=rank1, hooks=ctx.hooks)
            if rank1:
                dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xc, W1, W2, Ppre, dPl_c, None, dX=dX, accumulate_dX=True,
                                                           workspace=pws, ws_from_fwd=True,
                                                           ext_rank1=(dZ, Wa.reshape(-1)))
            else:
                dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xc, W1, W2, Ppre, dPl_c, dZ, dX=dX, accumulate_dX=True,
                                                           workspace=pws, ws_from_fwd=True)
        else:
            dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xc, W1, W2, Ppre, dPl_c, None, workspace=pws,
                                                       ws_from_fwd=True)
        return (dX.view(ctx.xshape), dW1, db1, dW2, db2, dWa, dba, dWt, dbt, None, None, None, None, None, None)


def attentional_pooling(X, Xatt, Wa, ba, Wt, bt, *, softmax_att=False, relu_att=False,
                        is_training=False, keep_prob=0.2, seed=0, offset=0, want_topdown=False,
                        relu_input=False, hooks=None):
    """`relu_input=True`: X is the backbone's pre-activation map; the op computes with max(X, 0) and
    returns the gradient w.r.t. the pre-activation (APA_FLAG_RELU_INPUT, include/apa.h)."""
    flags = cof.attn_flags(softmax_att, relu_att, is_training, relu_input)
    return AttentionalPoolingFunction.apply(X, Xatt, Wa, ba, Wt, bt, flags,
                                            keep_prob if is_training else 1.0, seed, offset,
                                            want_topdown, hooks)


class AttentionalPoolingHead(nn.Module):
    """PoseLogits head + the `USE_POSE_PRELOGITS_BASED_ATTENTION` head (nets_factory.py:147-160,
    242-352) as a module.

    Parameters (attribute -> TF variable name), reference initialisers in brackets:
      pose_w1 [C,768], pose_b1      PoseLogits/ExtraConv2d_1x1/{weights,biases}   [N(0,1e-3), 0]
      pose_w2 [768,J], pose_b2      PoseLogits/Conv2d_1c_1x1/{weights,biases}     [variance scaling, 0]
      att_weights [Cin,M], att_biases   PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/{..}  [N(0,1e-3), 0]
      td_weights [C,K], td_biases       PosePrelogitsBasedAttention/Conv/{..}                  [N(0,1e-3), 0]
    The pose head is only *evaluated* when something consumes it (attention from pose_pre_logits,
    cfg 003, or `with_pose_logits=True` for the pose loss) -- TF prunes it likewise.
    """

    TF_NAMES = {
        'pose_w1': 'PoseLogits/ExtraConv2d_1x1/weights', 'pose_b1': 'PoseLogits/ExtraConv2d_1x1/biases',
        'pose_w2': 'PoseLogits/Conv2d_1c_1x1/weights', 'pose_b2': 'PoseLogits/Conv2d_1c_1x1/biases',
        'att_weights': 'PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/weights',
        'att_biases': 'PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/biases',
        'td_weights': 'PosePrelogitsBasedAttention/Conv/weights',
        'td_biases': 'PosePrelogitsBasedAttention/Conv/biases',
    }
    POSE_PRELOGITS = 768

    # what the backbone's arg_scope makes of the un-annotated ..._WITH_POSE_FEAT_2LAYER conv (nets_factory.py:
    # 291-294 passes neither activation_fn nor normalizer_fn): (batch-norm?, eps, gamma?, relu?) -- executed from
    # the reference's own arg-scope functions in tests/golden/make_head_reference.py
    ARG_SCOPES = {
        'resnet': dict(bn=True, eps=1e-5, decay=0.997, scale=True, relu=True),          # resnet_utils.py:232-246
        'inception_v3': dict(bn=True, eps=1e-3, decay=0.9997, scale=False, relu=True),  # inception_utils.py:48-71
        'vgg': dict(bn=False, relu=True),                                               # vgg.py:58-62
        'inception_v2_tsn': dict(bn=False, relu=False),                                 # inception_v2_tsn.py:320-329
    }

    def __init__(self, num_classes: int, cfg, in_channels: int = 2048, num_pose_keypoints: int = 16,
                 is_training: bool = False, seed: int = 42, with_pose_logits: Optional[bool] = None,
                 want_topdown: bool = False, fuse_pose_attention: bool = True,
                 pose_in_channels: Optional[int] = None, arg_scope: str = 'resnet'):
        super().__init__()
        if arg_scope not in self.ARG_SCOPES:
            raise ValueError('arg_scope must be one of %s' % sorted(self.ARG_SCOPES))
        self.arg_scope = arg_scope
        self._replay_mask = None      # replay_dropout_mask(): an externally drawn keep mask for the next forward
        # apa_hooks (cof.make_hooks / deploy.OverlappedGradientSum.hooks) handed to every pooling call of this
        # head, forward and backward, also when autograd runs the backward on its engine thread
        self.hooks = None
        self.fuse_pose_attention = fuse_pose_attention   # cfg 003: one autograd node for pose head + pooling
        net = cfg.NET
        if not net.USE_POSE_PRELOGITS_BASED_ATTENTION:
            raise ValueError('AttentionalPoolingHead needs cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION')
        self.rank = int(net.USE_POSE_PRELOGITS_BASED_ATTENTION_RANK)
        self.with_pose_feat = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT)
        if self.rank < 1:
            raise ValueError('USE_POSE_PRELOGITS_BASED_ATTENTION_RANK must be >= 1')
        if self.rank > 1 and net.USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT:
            # does not build in the reference either: a 4-entry transpose perm on the stacked 5-D tensor
            # (nets_factory.py:271-279)
            raise ValueError('USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT with ..._RANK > 1 is not a valid '
                             'reference configuration (nets_factory.py:271-279)')
        self.pose_feat_2layer = bool(self.with_pose_feat and
                                     net.USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT_2LAYER)
        # rank > 1: identity activation on ONE map collapses to two streaming passes for any R (forward());
        # relu / per-class maps / the TopDownAttention dump / the pose features run one pass of the op per rank
        self.rank_collapsed = bool(self.rank > 1 and not (
            net.USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS or net.USE_POSE_PRELOGITS_BASED_ATTENTION_RELU_ATT
            or want_topdown or self.with_pose_feat))
        self.num_classes = num_classes
        self.single_layer = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT)
        self.softmax_att = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT)
        self.relu_att = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_RELU_ATT)
        self.per_class = bool(net.USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS)
        self.keep_prob = dropout_keep_prob(cfg)
        self.is_training = is_training
        self.seed = seed
        self.want_topdown = want_topdown
        if with_pose_logits is None:  # the pose logits are needed when a pose loss is configured
            with_pose_logits = bool(is_training and cfg.TRAIN.LOSS_FN_POSE and num_pose_keypoints > 0)
        self.with_pose_logits = with_pose_logits or self.with_pose_feat
        self._step = 0
        n_maps = num_classes if self.per_class else 1
        cp = self.POSE_PRELOGITS
        cin = in_channels if self.single_layer else cp
        # cfg.NET.LAST_CONV_MAP_FOR_POSE may name another end point with its own channel count (:148-150)
        self.pose_w1 = nn.Parameter(torch.randn(pose_in_channels or in_channels, cp) * 0.001)
        self.pose_b1 = nn.Parameter(torch.zeros(cp))
        # slim.variance_scaling_initializer(): truncated normal, std = sqrt(1.3 * 2 / fan_in)
        self.pose_w2 = nn.Parameter(torch.nn.init.trunc_normal_(
            torch.empty(cp, max(num_pose_keypoints, 1)), 0.0, 1.0, -2.0, 2.0) * (2.6 / cp) ** 0.5)
        self.pose_b2 = nn.Parameter(torch.zeros(max(num_pose_keypoints, 1)))
        self.att_weights = nn.Parameter(torch.randn(cin, n_maps) * 0.001)
        self.att_biases = nn.Parameter(torch.zeros(n_maps))
        # ..._WITH_POSE_FEAT: the top-down conv sees concat(last_conv, pose_logits) (:289-295)
        td_in = in_channels + (max(num_pose_keypoints, 1) if self.with_pose_feat else 0)
        self.in_channels = in_channels
        self.td_weights = nn.Parameter(torch.randn(td_in, num_classes) * 0.001)
        self.td_biases = nn.Parameter(torch.zeros(num_classes))
        if self.pose_feat_2layer:
            # :290-294: slim.conv2d(pose_logits, J, [1,1], N(0,1e-3)) with neither normalizer_fn nor
            # activation_fn given, so the resnet arg-scope's batch-norm (decay 0.997, eps 1e-5, scale)
            # + relu apply and there is no bias; the batch-norm sits OUTSIDE resnet_v1()'s own
            # `arg_scope([slim.batch_norm], is_training=...)`, so it keeps slim's default
            # is_training=True: batch statistics in training AND in evaluation (reproduced literally).
            # Unnamed conv created before the top-down conv -> it is scope 'Conv' and the top-down
            # conv becomes 'Conv_1' (tf_variable_names()).  [N,P,J]-sized: torch ops on the device.
            # Under the other backbones' arg-scopes (ARG_SCOPES) the same line builds a different layer:
            # vgg: biases + relu, no normalizer; inception_v2_tsn: biases, linear; inception_v3: batch-norm
            # without gamma (eps 1e-3) + relu.
            J = max(num_pose_keypoints, 1)
            sc = self.ARG_SCOPES[arg_scope]
            self.pose_feat_weights = nn.Parameter(torch.randn(J, J) * 0.001)
            if sc['bn']:
                if sc['scale']:
                    self.pose_feat_bn_gamma = nn.Parameter(torch.ones(J))
                self.pose_feat_bn_beta = nn.Parameter(torch.zeros(J))
                self.register_buffer('pose_feat_bn_moving_mean', torch.zeros(J))
                self.register_buffer('pose_feat_bn_moving_variance', torch.ones(J))
            else:
                self.pose_feat_biases = nn.Parameter(torch.zeros(J))
        # rank > 1 (:258-274, :298-309): the r-th attention conv consumes the OUTPUT of conv r-1
        # ([M,M] = [1,1] weights, scopes Conv2d_PrePose_Attn1, ...); one more top-down conv per rank
        # (scopes Conv_1, ...)
        self.att_weights_r = nn.ParameterList(
            [nn.Parameter(torch.randn(n_maps, n_maps) * 0.001) for _ in range(self.rank - 1)])
        self.att_biases_r = nn.ParameterList(
            [nn.Parameter(torch.zeros(n_maps)) for _ in range(self.rank - 1)])
        self.td_weights_r = nn.ParameterList(
            [nn.Parameter(torch.randn(td_in, num_classes) * 0.001) for _ in range(self.rank - 1)])
        self.td_biases_r = nn.ParameterList(
            [nn.Parameter(torch.zeros(num_classes)) for _ in range(self.rank - 1)])

    def tf_variable_names(self):
        """attribute -> TF variable name, including the per-rank scopes."""
        names = dict(self.TF_NAMES)
        if getattr(self, 'pose_feat_2layer', False):
            pre = 'PosePrelogitsBasedAttention/'
            names.update({'pose_feat_weights': pre + 'Conv/weights',
                          'td_weights': pre + 'Conv_1/weights', 'td_biases': pre + 'Conv_1/biases'})
            sc = self.ARG_SCOPES[self.arg_scope]
            if sc['bn']:
                names.update({'pose_feat_bn_beta': pre + 'Conv/BatchNorm/beta',
                              'pose_feat_bn_moving_mean': pre + 'Conv/BatchNorm/moving_mean',
                              'pose_feat_bn_moving_variance': pre + 'Conv/BatchNorm/moving_variance'})
                if sc['scale']:
                    names['pose_feat_bn_gamma'] = pre + 'Conv/BatchNorm/gamma'
            else:
                names['pose_feat_biases'] = pre + 'Conv/biases'
        two = 1 if getattr(self, 'pose_feat_2layer', False) else 0
        for r in range(1, self.rank):
            # unnamed convs are numbered in creation order: the _2LAYER conv (if any) took 'Conv'
            pre = 'PosePrelogitsBasedAttention/'
            names['att_weights_r.%d' % (r - 1)] = pre + 'Conv2d_PrePose_Attn%d/weights' % r
            names['att_biases_r.%d' % (r - 1)] = pre + 'Conv2d_PrePose_Attn%d/biases' % r
            names['td_weights_r.%d' % (r - 1)] = pre + 'Conv_%d/weights' % (r + two)
            names['td_biases_r.%d' % (r - 1)] = pre + 'Conv_%d/biases' % (r + two)
        return names

    def replay_dropout_mask(self, keep_mask: Optional[torch.Tensor]) -> None:
        """Use an externally drawn dropout mask for the NEXT training-mode forward (and its backward) instead of
        the head's own counter-hash stream: `keep_mask` is the {0,1} tensor tf.nn.dropout multiplies the top-down
        input with (floor(keep_prob + U), nets_factory.py:296), shaped like that input -- [N,H,W,C], or
        [N,H,W,C+J] with ..._WITH_POSE_FEAT.  One-shot; None clears it.  (APA_FLAG_RNG_EXTERNAL, include/apa.h:
        a parity facility -- the replayed mask runs through the library's generic kernels.)"""
        self._replay_mask = keep_mask

    def _dropout_seed(self, split_at: Optional[int] = None):
        """the `seed` argument of the pooling ops for this forward: the head's integer seed, or the packed replay
        mask (split_at = C: the split-channel *_cat ops take X's elements first, then the extra channels)"""
        m = self._replay_mask
        if m is None or not self.is_training:
            return self.seed
        dev = self.td_weights.device
        if split_at is None:
            return cof.pack_keep_mask(m, device=dev)
        return cof.pack_keep_mask(m[..., :split_at].contiguous(), m[..., split_at:].contiguous(), device=dev)

    def get_extra_state(self):
        # the dropout step counter keys the mask stream: checkpoint it, so a resumed run does not
        # replay the masks of steps 0, 1, ...
        return {'dropout_step': int(self._step)}

    def set_extra_state(self, state):
        self._step = int(state.get('dropout_step', 0)) if state else 0

    def regularized_weights(self):
        """conv weights carry slim.l2_regularizer from the backbone's arg-scope (resnet_utils.py:241);
        biases and batch-norm parameters do not."""
        # The PoseLogits convs are ALWAYS built (:147-160) and their weights sit in REGULARIZATION_LOSSES even
        # when nothing consumes the pose head (cfg 002: TF prunes the ops, not the regulariser) -- the reference's
        # total loss and the decay of those weights include them (tests/golden/ref_head_cfg002_*.npz).
        ws = [self.pose_w1, self.pose_w2, self.att_weights, self.td_weights] + list(self.att_weights_r) + \
            list(self.td_weights_r)
        if getattr(self, 'pose_feat_2layer', False):
            ws.append(self.pose_feat_weights)
        return ws

    def can_fuse_input_relu(self, dtype=torch.float32) -> bool:
        """True when the conv5 map has no consumer but the rank-1, single-layer attention op of the
        streaming kernels -- then the backbone may hand over its PRE-activation map and the op applies
        the final ReLU on the fly (forward(..., preactivation=True))."""
        c_ok = self.in_channels in ((1024, 2048, 4096) if dtype == torch.float32 else (2048,))
        return (self.rank == 1 and self.single_layer and not self.per_class and not self.with_pose_logits
                and not self.with_pose_feat and not self.want_topdown and c_ok)

    def forward(self, last_conv: torch.Tensor, preactivation: bool = False,
                last_conv_pose: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """`last_conv_pose`: the pose head's own feature tap when cfg.NET.LAST_CONV_MAP_FOR_POSE names a
        different end point than last_conv_map (nets_factory.py:148-150; inception_v2_tsn: 5a vs 5b).
        None / the same tensor: the shared tap of the ResNet configs."""
        replay = self._replay_mask is not None and self.is_training
        if preactivation and (replay or not self.can_fuse_input_relu(last_conv.dtype)):
            last_conv, preactivation = torch.relu(last_conv), False
        if last_conv_pose is None:
            last_conv_pose = last_conv
        end_points: Dict[str, torch.Tensor] = EndPoints()
        pose_pre = None
        if (not self.single_layer and self.rank == 1 and not self.per_class and not self.with_pose_feat
                and self.fuse_pose_attention and last_conv_pose is last_conv):
            # cfg 003: pose head + attention on its pre-logits as one autograd node (rank-1 hand-over)
            offset = self._step
            if self.is_training:
                self._step += 1
            n, h, w = last_conv.shape[0], last_conv.shape[1], last_conv.shape[2]
            flags = cof.attn_flags(self.softmax_att, self.relu_att, self.is_training)
            logits, att, pose_logits, topdown = PoseAttentionFunction.apply(
                last_conv, self.pose_w1, self.pose_b1, self.pose_w2, self.pose_b2, self.att_weights,
                self.att_biases, self.td_weights, self.td_biases, flags,
                self.keep_prob if self.is_training else 1.0, self._dropout_seed(), offset, self.want_topdown,
                self.hooks)
            self._replay_mask = None
            end_points['PoseLogits'] = pose_logits
            end_points['PosePrelogitsBasedAttention'] = att.view(n, h, w, -1)
            if topdown is not None:
                end_points['TopDownAttention'] = topdown.view(n, h, w, -1)
            end_points['Logits'] = logits
            return logits, end_points
        if self.with_pose_logits or not self.single_layer:          # :147-160
            pose_pre, pose_logits = PoseHeadFunction.apply(last_conv_pose, self.pose_w1, self.pose_b1,
                                                           self.pose_w2, self.pose_b2)
            end_points['PoseLogits'] = pose_logits
        else:   # built by the reference all the same (:147-160), evaluated only if fetched
            # With the fused input ReLU the shared tap is the backbone's PRE-activation sum: the reference's
            # PoseLogits convs read the rectified block4 end point (train.py:399-400 fetches it whatever
            # the head does with it), so rectify inside the thunk -- paid only by a caller that fetches it.
            pose_needs_relu = preactivation and last_conv_pose is last_conv
            pose_src = last_conv_pose
            end_points.lazy('PoseLogits', lambda: PoseHeadFunction.apply(
                torch.relu(pose_src) if pose_needs_relu else pose_src,
                self.pose_w1, self.pose_b1, self.pose_w2, self.pose_b2)[1])
        xatt = None if self.single_layer else pose_pre               # :247-250
        offset = self._step
        if self.is_training:
            self._step += 1            # a fresh dropout mask per step
        n, h, w = last_conv.shape[0], last_conv.shape[1], last_conv.shape[2]
        C = self.in_channels
        kw = dict(softmax_att=self.softmax_att, relu_att=self.relu_att, is_training=self.is_training,
                  keep_prob=self.keep_prob, seed=self.seed, offset=offset, hooks=self.hooks)    # seed: see below

        xext = None
        if self.with_pose_feat:
            # :289-296: concat(last_conv, pose_logits) -> dropout -> top-down conv.  The J extra channels
            # run through the same HIP op (apa_attn_pool_*_cat): same dropout stream, same attention-
            # weighted mean, their share of dA fed to the streaming backward kernel.
            xext = end_points['PoseLogits']
            if self.pose_feat_2layer:
                sc = self.ARG_SCOPES[self.arg_scope]
                y = xext.float() @ self.pose_feat_weights
                if sc['bn']:
                    mean = y.mean(dim=(0, 1, 2))
                    var = y.var(dim=(0, 1, 2), unbiased=False)           # tf.nn.moments
                    if self.is_training:
                        with torch.no_grad():                             # UPDATE_OPS: mov -= (1 - decay) (mov - batch)
                            d = 1.0 - sc['decay']
                            self.pose_feat_bn_moving_mean.mul_(1.0 - d).add_(mean.detach(), alpha=d)
                            self.pose_feat_bn_moving_variance.mul_(1.0 - d).add_(var.detach(), alpha=d)
                    inv = torch.rsqrt(var + sc['eps'])
                    if sc['scale']:
                        inv = inv * self.pose_feat_bn_gamma
                    y = (y - mean) * inv + self.pose_feat_bn_beta
                else:
                    y = y + self.pose_feat_biases
                xext = torch.relu(y) if sc['relu'] else y
        x_td, xatt_td, cat_op = last_conv, xatt, self.with_pose_feat
        if self.with_pose_feat and (self.per_class or self.want_topdown):
            # per-class maps + pose features, or the TopDownAttention dump of the pose-feature head (no shipped
            # config): the M == K kernels have no split-channel form and the split-channel M == 1 entry
            # points never form the top-down tensor, so the concatenation is made as the reference makes it
            # (tf.concat, :295 -- a copy, not arithmetic) and the plain op runs on it with the map itself as its
            # separate attention input; the dropout mask is then the flat stream over the [N,H,W,C+J] tensor
            # The kernels read whole 16-byte vectors: C + J is padded with zero channels (and the top-down weights
            # with zero rows) to the next multiple of 4 fp32 / 8 bf16 channels -- J = 13 keypoints work.  (The pose
            # logits ride in the feature dtype here: bf16 features round them to bf16, the split-channel M == 1 ops
            # keep them in fp32.)
            vec = 4 if last_conv.dtype == torch.float32 else 8
            td_pad = (-(C + xext.shape[-1])) % vec
            parts = [last_conv, xext.to(last_conv.dtype)]
            if td_pad:
                parts.append(last_conv.new_zeros(n, h, w, td_pad))
            x_td = torch.cat(parts, dim=-1)
            xatt_td = last_conv if self.single_layer else pose_pre
            cat_op = False
            if td_pad and self._replay_mask is not None:
                m_ = self._replay_mask
                self._replay_mask = torch.cat([m_, m_.new_zeros(tuple(m_.shape[:-1]) + (td_pad,))], dim=-1)
        else:
            td_pad = 0
        pad_rows = (lambda w_: torch.cat([w_, w_.new_zeros(td_pad, w_.shape[1])], dim=0)) if td_pad else (lambda w_: w_)
        # the dropout key of every pooling pass below: the head's own seed, or the one-shot replay mask
        seed = self._dropout_seed(split_at=C if (self.with_pose_feat and cat_op) else None)
        kw['seed'] = seed
        self._replay_mask = None
        if self.rank == 1 and cat_op:
            flags = cof.attn_flags(self.softmax_att, self.relu_att, self.is_training)
            logits, att = AttentionalPoolingCatFunction.apply(
                last_conv, xatt, xext, self.att_weights, self.att_biases, self.td_weights, self.td_biases,
                flags, self.keep_prob if self.is_training else 1.0, seed, offset, self.hooks)
            end_points['PosePrelogitsBasedAttention'] = att.view(n, h, w, -1)    # :287
        elif self.rank == 1:
            logits, att, topdown = attentional_pooling(
                x_td, xatt_td, self.att_weights, self.att_biases, pad_rows(self.td_weights), self.td_biases,
                want_topdown=self.want_topdown, relu_input=preactivation, **kw)
            end_points['PosePrelogitsBasedAttention'] = att.view(n, h, w, -1)    # :287
            if topdown is not None:
                end_points['TopDownAttention'] = topdown.view(n, h, w, -1)       # :309
        elif not self.rank_collapsed:
            # rank R > 1 in general (:258-274, :298-309, :322-328): conv r consumes the output of conv r-1,
            # so map r is an affine function of the attention input with the effective weights
            #   Wa_r = Wa_0 W_1 .. W_r,   ba_r = ba_{r-1} W_r + b_r          ([Cin,M] / [M], M = 1 or K)
            # (tiny products of PARAMETERS, differentiable torch ops); the activation acts on every map
            # separately and the ranks are summed after the spatial mean -- one pass of the HIP op per rank,
            # all with the same dropout mask (the reference drops last_conv once, before the rank loop).
            wa_r, ba_r = self.att_weights, self.att_biases
            wts = [self.td_weights] + list(self.td_weights_r)
            bts = [self.td_biases] + list(self.td_biases_r)
            logits, atts, tds = None, [], []
            for r in range(self.rank):
                if r > 0:
                    wa_r = wa_r @ self.att_weights_r[r - 1]
                    ba_r = ba_r @ self.att_weights_r[r - 1] + self.att_biases_r[r - 1]
                if cat_op:
                    flags = cof.attn_flags(self.softmax_att, self.relu_att, self.is_training)
                    lg_r, att_r = AttentionalPoolingCatFunction.apply(
                        last_conv, xatt, xext, wa_r, ba_r, wts[r], bts[r], flags,
                        self.keep_prob if self.is_training else 1.0, seed, offset, self.hooks)
                    td_r = None
                else:
                    lg_r, att_r, td_r = attentional_pooling(x_td, xatt_td, wa_r, ba_r, pad_rows(wts[r]), bts[r],
                                                            want_topdown=self.want_topdown, **kw)
                logits = lg_r if logits is None else logits + lg_r
                atts.append(att_r.view(n, h, w, -1))
                if td_r is not None:
                    tds.append(td_r.view(n, h, w, -1))
            end_points['PosePrelogitsBasedAttention'] = torch.stack(atts, dim=-1)   # [N,H,W,M,R] (:271-274)
            if tds:
                end_points['TopDownAttention'] = torch.stack(tds, dim=-1)           # [N,H,W,K,R] (:305-309)
        else:
            # rank R > 1, identity activation, one bottom-up map.  Z_r = Z_{r-1} w_r + b_r with scalar
            # (w_r, b_r), so Z_r = alpha_r Z_0 + beta_r and
            #   logits = sum_r mean_p Z_r (X' Wt_r + bt_r)
            #          = pool(X; Z_0, sum_r alpha_r Wt_r, sum_r alpha_r bt_r)          (pass A)
            #          + mean_p X' . sum_r beta_r Wt_r + sum_r beta_r bt_r             (pass B)
            # i.e. two streaming passes of the rank-1 HIP op for any R (pass B is the op with a
            # constant attention map), same dropout mask in both.
            one = torch.ones((), device=last_conv.device)
            alphas, betas = [one], [torch.zeros((), device=last_conv.device)]
            for r in range(self.rank - 1):
                wr, br = self.att_weights_r[r].reshape(()), self.att_biases_r[r].reshape(())
                alphas.append(alphas[-1] * wr)
                betas.append(betas[-1] * wr + br)
            wts = [self.td_weights] + list(self.td_weights_r)
            bts = [self.td_biases] + list(self.td_biases_r)
            wt_a = sum(a * w_ for a, w_ in zip(alphas, wts))
            bt_a = sum(a * b_ for a, b_ in zip(alphas, bts))
            wt_b = sum(b * w_ for b, w_ in zip(betas, wts))
            c_b = sum(b * b_ for b, b_ in zip(betas, bts))
            logits_a, att, _ = attentional_pooling(last_conv, xatt, self.att_weights, self.att_biases,
                                                   wt_a, bt_a, **kw)
            ones_in = torch.zeros(n, h, w, 8, device=last_conv.device, dtype=last_conv.dtype)
            logits_b, _, _ = attentional_pooling(
                last_conv, ones_in, torch.zeros(8, 1, device=last_conv.device),
                torch.ones(1, device=last_conv.device), wt_b,
                torch.zeros_like(self.td_biases), **kw)
            logits = logits_a + logits_b + c_b
            z0 = att.view(n, h, w, 1)
            end_points['PosePrelogitsBasedAttention'] = torch.stack(
                [a * z0 + b for a, b in zip(alphas, betas)], dim=-1)            # [N,H,W,1,R] (:271-274)
        end_points['Logits'] = logits                                            # :352
        return logits, end_points


class SpatialMeanFunction(torch.autograd.Function):
    """z [N,C] f32 = mean over the spatial positions of X [N,H,W,C] (resnet_v1.py:206-208), in HIP both ways:
    forward = the pooling pass with a constant attention map (one streaming read of X; its `zsave` output),
    backward = apa_spatial_mean_bwd (one streaming write)."""

    @staticmethod
    def forward(ctx, X):
        n, C = X.shape[0], X.shape[-1]
        dev = X.device
        ones_in = torch.zeros(tuple(X.shape[:-1]) + (8,), device=dev, dtype=X.dtype)
        _, _, z, _, _, _ = cof.attn_pool_fwd(X.contiguous(), ones_in, torch.zeros(8, 1, device=dev),
                                             torch.ones(1, device=dev), torch.zeros(C, 1, device=dev),
                                             torch.zeros(1, device=dev))
        ctx.x_shape, ctx.x_dtype = tuple(X.shape), X.dtype
        return z

    @staticmethod
    def backward(ctx, dz):
        return cof.spatial_mean_bwd(dz.contiguous().float(), ctx.x_shape, ctx.x_dtype)


class BaselineHead(nn.Module):
    """cfg 001 (`experiments/001_MPII_ResNet.yaml`, no attention): the slim ResNet's own head --
    global average pool, dropout, 1x1 conv `resnet_v1_101/logits` (models/slim/nets/resnet_v1.py:206-217; the
    dropout exists only when cfg.NET.DROPOUT >= 0, nets_factory.py:127-129).  Not the hot path: it is
    the plumbing baseline of BASELINE configs[0].  In eval mode it IS the attentional-pooling op with
    a constant attention map (mean_p X . W + b, one streaming pass in HIP).  In training the dropout
    sits between the pooled vector and the classifier: the pooled vector z comes from SpatialMeanFunction,
    and dropout + classifier are the pooling op again on z seen as a 1 x 1 map (P = 1: its dropout on the
    "features" is the dropout on z, same counter-based mask stream, `cof.dropout_mask((N, C), ...)`)."""

    TF_NAMES = {'logits_weights': 'logits/weights', 'logits_biases': 'logits/biases',
                'pose_w1': 'PoseLogits/ExtraConv2d_1x1/weights', 'pose_b1': 'PoseLogits/ExtraConv2d_1x1/biases',
                'pose_w2': 'PoseLogits/Conv2d_1c_1x1/weights', 'pose_b2': 'PoseLogits/Conv2d_1c_1x1/biases'}

    def __init__(self, num_classes: int, cfg, in_channels: int = 2048, is_training: bool = False, seed: int = 42,
                 num_pose_keypoints: int = 16):
        super().__init__()
        self.num_classes = num_classes
        # The PoseLogits convs are built in EVERY configuration (nets_factory.py:147-160), this one included: nothing
        # consumes them here, but they are variables of the graph (checkpoints carry them) and their weights sit in
        # REGULARIZATION_LOSSES -- the reference's total loss and weight decay include them
        # (tests/golden/ref_e2e.npz: reg_groups['PoseLogits']).  Same initialisers as AttentionalPoolingHead.
        cp = AttentionalPoolingHead.POSE_PRELOGITS
        self.pose_w1 = nn.Parameter(torch.randn(in_channels, cp) * 0.001)
        self.pose_b1 = nn.Parameter(torch.zeros(cp))
        self.pose_w2 = nn.Parameter(torch.nn.init.trunc_normal_(
            torch.empty(cp, max(num_pose_keypoints, 1)), 0.0, 1.0, -2.0, 2.0) * (2.6 / cp) ** 0.5)
        self.pose_b2 = nn.Parameter(torch.zeros(max(num_pose_keypoints, 1)))
        # The backbone's own dropout is only configured when cfg.NET.DROPOUT >= 0 (nets_factory.py:127-129:
        # `kwargs['dropout_keep_prob'] = 1 - DROPOUT`); with the default DROPOUT = -1 -- the shipped cfg 001 --
        # nothing is passed and resnet_v1_101's default keep probability 1.0 applies: NO dropout.  (The 0.2 rule
        # of :143-146 is an arg-scope for the slim.dropout calls of the attention head only.)
        self.keep_prob = 1.0 if cfg.NET.DROPOUT < 0 else 1.0 - float(cfg.NET.DROPOUT)
        self.is_training = is_training
        self.seed = seed
        self._step = 0
        # slim.variance_scaling_initializer(): truncated normal, std = sqrt(1.3 * 2 / fan_in)
        self.logits_weights = nn.Parameter(torch.nn.init.trunc_normal_(
            torch.empty(in_channels, num_classes), 0.0, 1.0, -2.0, 2.0) * (2.6 / in_channels) ** 0.5)
        self.logits_biases = nn.Parameter(torch.zeros(num_classes))

    def get_extra_state(self):
        return {'dropout_step': int(self._step)}

    def set_extra_state(self, state):
        self._step = int(state.get('dropout_step', 0)) if state else 0

    def regularized_weights(self):
        return [self.logits_weights, self.pose_w1, self.pose_w2]

    def forward(self, last_conv: torch.Tensor):
        n, h, w = last_conv.shape[0], last_conv.shape[1], last_conv.shape[2]
        if self.is_training and self.keep_prob < 1.0:
            z = SpatialMeanFunction.apply(last_conv)                    # [N, C] f32
            offset = self._step
            self._step += 1            # a fresh dropout mask per step
            dev = z.device
            logits, _, _ = attentional_pooling(
                z.view(n, 1, 1, -1), torch.zeros(n, 1, 1, 8, device=dev), torch.zeros(8, 1, device=dev),
                torch.ones(1, device=dev), self.logits_weights, self.logits_biases, is_training=True,
                keep_prob=self.keep_prob, seed=self.seed, offset=offset)
        else:
            ones_in = torch.zeros(n, h, w, 8, device=last_conv.device, dtype=last_conv.dtype)
            logits, _, _ = attentional_pooling(
                last_conv, ones_in, torch.zeros(8, 1, device=last_conv.device),
                torch.ones(1, device=last_conv.device), self.logits_weights, self.logits_biases)
        return logits, {'Logits': logits}


class FramePoolFunction(torch.autograd.Function):
    """[B*F,K] -> [B,K]: mean over frames, optionally weighted by the temporal attention
    a = logits.w + b (nets_factory.py:354-374), forward and backward in HIP."""

    @staticmethod
    def forward(ctx, logits, w, b, frames_per_video):
        lc = logits.contiguous().float()
        pooled, tatt = cof.frame_pool_fwd(lc, frames_per_video,
                                          None if w is None else w.contiguous().view(-1),
                                          None if b is None else b.contiguous().view(-1))
        ctx.fpv = frames_per_video
        ctx.has_att = w is not None
        ctx.save_for_backward(lc, *( [w.contiguous().view(-1), tatt] if w is not None else [] ))
        if tatt is not None:
            ctx.mark_non_differentiable(tatt)
        return pooled, tatt

    @staticmethod
    def backward(ctx, dpooled, _dtatt):
        saved = ctx.saved_tensors
        lc = saved[0]
        w, tatt = (saved[1], saved[2]) if ctx.has_att else (None, None)
        dlogits, dw, db = cof.frame_pool_bwd(lc, ctx.fpv, w, tatt, dpooled.contiguous().float())
        if dw is not None:
            dw = dw.view(-1, 1)
        return dlogits, dw, db, None


def frame_pooling(logits: torch.Tensor, frames_per_video: int, end_points: Dict[str, torch.Tensor],
                  temporal_w: Optional[torch.Tensor] = None, temporal_b: Optional[torch.Tensor] = None
                  ) -> torch.Tensor:
    """nets_factory.py:354-374: end_points['logits_beforePool'], optional temporal attention
    (end_points['TemporalAttention'], [B,F,1,1] like the reference's conv output), mean over F."""
    end_points['logits_beforePool'] = logits
    pooled, tatt = FramePoolFunction.apply(logits, temporal_w, temporal_b, frames_per_video)
    if tatt is not None:
        end_points['TemporalAttention'] = tatt.view(-1, frames_per_video, 1, 1)
    return pooled


def get_network_fn(name: str, num_classes: int, num_pose_keypoints: int, cfg,
                   weight_decay: float = 0.0, is_training: bool = False,
                   backbone: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                   device='cuda', with_backbone: bool = False, backbone_dtype=None,
                   fuse_final_relu: bool = False, clone_index: Optional[int] = None, **head_kwargs):
    """Same signature as nets_factory.py:94-95 (+ optional backbone/device).

    Returns `network_fn(images) -> (logits, end_points)`; `network_fn.head` exposes the module
    (its parameters) and `network_fn.weight_decay` the coefficient for the L2 term.
    With `backbone=None`, `images` IS the conv5 map [N,H,W,C] (or [B,F,H,W,C] for video input,
    nets_factory.py:121-125).  `with_backbone=True` builds the slim ResNet-v1 of `resnet_v1.py`
    (PyTorch-ROCm, channels-last: its block4 tap is read by the HIP op with no layout change);
    `backbone_dtype=torch.bfloat16` runs it under autocast so the tap arrives as bf16.
    `fuse_final_relu=True` (with the built-in backbone): block4 hands over its residual sum BEFORE the
    last ReLU and the pooling op applies it on the fly in both passes (the ReLU's
    own read + write of the map and its backward pass never run); used whenever the head is the map's
    only consumer (`head.can_fuse_input_relu`), otherwise the ReLU is applied explicitly.
    """
    if name not in last_conv_map:
        raise ValueError('Name of network unknown %s' % name)
    # `in_channels=` overrides the tap's nominal channel count (a caller-supplied backbone / feature map)
    channels = head_kwargs.pop('in_channels', None) or last_conv_map[name][1]
    head_kwargs.setdefault('arg_scope', ARG_SCOPE_OF[name])
    if clone_index is None:      # one clone per data-parallel rank (model_deploy.py:189-197)
        import torch.distributed as dist
        clone_index = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    # the reference's towers draw independent dropout masks: fold the clone index into the key
    head_seed = int(cfg.RNG_SEED) + 1000003 * int(clone_index)
    if with_backbone and backbone is None:
        from . import resnet_v1
        if name not in resnet_v1.BLOCKS:
            raise ValueError('no built-in backbone for %s (pass backbone=callable)' % name)
        net = resnet_v1.ResNetV1(name, final_relu=not fuse_final_relu).to(device)
        net.train(is_training)
        if cfg.NET.TRAIN_TOP_BN:
            resnet_v1.freeze_all_but_root_batch_norm(net)      # resnet_v1.py:191-204

        def backbone(images, _net=net, _dt=backbone_dtype):
            if _dt is None:
                return _net(images)
            with torch.autocast('cuda', dtype=_dt):
                return _net(images)
        backbone.module = net
    else:
        fuse_final_relu = False
    if cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION:
        head = AttentionalPoolingHead(num_classes, cfg, in_channels=channels,
                                      num_pose_keypoints=num_pose_keypoints, is_training=is_training,
                                      seed=head_seed, **head_kwargs).to(device)
    else:   # cfg 001: the backbone's own average-pool + logits head
        head_kwargs.pop('arg_scope', None)
        head = BaselineHead(num_classes, cfg, in_channels=channels, is_training=is_training,
                            seed=head_seed, num_pose_keypoints=num_pose_keypoints).to(device)
    temporal = None
    if cfg.NET.USE_TEMPORAL_ATT:
        # 'TemporalAttention/Conv/{weights,biases}': 1x1 conv K->1, N(0,1e-3) weights; the bias is
        # initialised to 1/frames_per_video at the first call (:366-368), when F is known
        temporal = nn.ParameterDict({
            'weights': nn.Parameter(torch.randn(num_classes, 1) * 0.001),
            'biases': nn.Parameter(torch.zeros(1))}).to(device)
        temporal._bias_initialised = False

    def network_fn(images: torch.Tensor):
        frames_per_video = 1
        if images.dim() == 5:                                   # :121-125
            frames_per_video = images.shape[1]
            images = images.reshape(-1, *images.shape[2:])
        last_conv = backbone(images) if backbone is not None else images
        if isinstance(last_conv, dict):
            # a backbone that returns its end points (slim style): the two taps by name (:136-140, :148-150)
            eps = last_conv
            last_conv = eps[last_conv_map[name][0]]
            pose_tap = eps.get(getattr(cfg.NET.LAST_CONV_MAP_FOR_POSE, name, None), last_conv)
            if isinstance(head, AttentionalPoolingHead) and pose_tap is not last_conv:
                logits, end_points = head(last_conv, last_conv_pose=pose_tap)
            else:
                logits, end_points = head(last_conv)
        elif fuse_final_relu and isinstance(head, AttentionalPoolingHead):
            logits, end_points = head(last_conv, preactivation=True)
        elif fuse_final_relu:
            logits, end_points = head(torch.relu(last_conv))
        else:
            logits, end_points = head(last_conv)
        if frames_per_video > 1:                                # :354-374
            tw = tb = None
            if temporal is not None:
                if not temporal._bias_initialised:
                    with torch.no_grad():
                        temporal['biases'].fill_(1.0 / frames_per_video)
                    temporal._bias_initialised = True
                tw, tb = temporal['weights'], temporal['biases']
            logits = frame_pooling(logits, frames_per_video, end_points, tw, tb)
        return logits, end_points

    def regularized_weights():
        """every tensor slim's arg-scope puts the L2 regulariser on (resnet_utils.py:241): the head's conv
        weights and the TemporalAttention conv weights (biases are not regularised)."""
        ws = list(head.regularized_weights())
        if temporal is not None:
            ws.append(temporal['weights'])
        return ws

    def features(images: torch.Tensor):
        """the backbone half of network_fn: images -> the conv5 map the head reads (the images themselves when no
        backbone was given) and whether it is the block's PRE-activation sum (fuse_final_relu) -- for callers that
        run the head themselves (deploy.FusedHeadStep: the head's forward, losses and backward as one host call)"""
        last_conv = backbone(images) if backbone is not None else images
        if isinstance(last_conv, dict):
            last_conv = last_conv[last_conv_map[name][0]]
        return last_conv, bool(fuse_final_relu)

    network_fn.features = features
    network_fn.head = head
    network_fn.regularized_weights = regularized_weights
    network_fn.backbone = getattr(backbone, 'module', backbone)
    network_fn.temporal = temporal
    network_fn.weight_decay = weight_decay
    network_fn.num_pose_keypoints = num_pose_keypoints
    return network_fn
