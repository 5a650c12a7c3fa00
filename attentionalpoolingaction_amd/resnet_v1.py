"""slim ResNet-v1 backbone on PyTorch-ROCm, emitting the conv5 tap the pooling op reads in place.

SURVEY.md section 8(f) row 1.  The backbone stays a library network (MIOpen convolutions through
torch): what matters for the hot path is the *boundary* -- the block4 output leaves here as a
channels-last tensor whose `[N,H,W,C]` view is contiguous, so `apa_attn_pool_fwd` consumes it with
zero layout change (and as bf16 when the backbone runs under autocast).

Semantics follow the reference's slim code, not torchvision's:
  * `conv2d_same` (models/slim/nets/resnet_utils.py:77-122): stride 1 -> SAME; stride > 1 ->
    explicit symmetric zero padding (k-1)//2 | k-1-(k-1)//2, then VALID.  For the odd kernels used
    here that is exactly `nn.Conv2d(padding=k//2)`.
  * `pool1` is `max_pool2d(3, stride 2, padding='SAME')` (resnet_utils.py:246-254): TF SAME pads
    `max((ceil(H/2)-1)*2 + 3 - H, 0)` split low//2 | rest, i.e. (0,1) for even H, (1,1) for odd --
    not torch's symmetric padding=1.
  * bottleneck (models/slim/nets/resnet_v1.py:68-112): the stride sits on the 3x3 conv of the LAST
    unit of blocks 1-3; identity shortcuts are `subsample` (= x[:, ::s, ::s]); every conv is
    followed by batch norm (decay 0.997, eps 1e-5, scale) and, except conv3 / shortcut, ReLU.
  * spatial sizes: 450 -> 225 -> 113 -> 57 -> 29 -> 15, 448 -> 14, 224 -> 7.
The known-answer tests of the reference (models/slim/nets/resnet_v1_test.py:58-152) are reproduced
in tests/test_resnet_cpu.py.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_DECAY, BN_EPS = 0.997, 1e-5      # resnet_utils.py:209-212


def subsample(x: torch.Tensor, factor: int) -> torch.Tensor:
    """resnet_utils.py:58-74: max_pool2d([1,1], stride=factor) == strided slicing.  x is NCHW."""
    return x if factor == 1 else x[:, :, ::factor, ::factor]


def tf_same_pad(size: int, k: int, s: int) -> Tuple[int, int]:
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def max_pool_same(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    pt, pb = tf_same_pad(x.shape[2], k, s)
    pl, pr = tf_same_pad(x.shape[3], k, s)
    if pt or pb or pl or pr:
        x = F.pad(x, (pl, pr, pt, pb), value=float('-inf'))
    return F.max_pool2d(x, k, s)


class SlimBatchNorm2d(nn.BatchNorm2d):
    """slim.batch_norm's moving statistics: the moving VARIANCE is updated with the batch variance exactly as
    tf.nn.moments returns it (biased, / n) -- nn.BatchNorm2d feeds the unbiased estimate (/ (n - 1)) into
    running_var.  The normalisation itself is identical; the update is corrected after the fact (a few
    C-sized operations, no extra pass over the activations).  Found by running the reference's graph code
    (tests/golden/make_backbone_reference.py)."""

    def forward(self, x):
        if not (self.training and self.track_running_stats) or self.momentum is None:
            return super().forward(x)
        n = x.numel() // x.shape[1]
        # the op updates (and autograd keeps a reference to) a COPY of the moving variance; the module's buffer
        # then receives the corrected update -- editing the buffer the op saw in place would invalidate the
        # tensor the backward pass saved
        seen = self.running_var.clone()
        y = F.batch_norm(x, self.running_mean, seen, self.weight, self.bias, True, self.momentum, self.eps)
        with torch.no_grad():
            old = self.running_var * (1.0 - self.momentum)
            unbias = (n - 1.0) / n if n > 1 else 1.0
            self.running_var.copy_((seen - old) * unbias + old)
            if self.num_batches_tracked is not None:
                self.num_batches_tracked += 1
        return y


def freeze_all_but_root_batch_norm(net: 'ResNetV1') -> None:
    """cfg.NET.TRAIN_TOP_BN (models/slim/nets/resnet_v1.py:191-204): every batch norm but the root block's runs
    with is_training=False and trainable=False -- moving statistics, frozen gamma / beta -- also through later
    `.train()` calls."""
    root_bn = net.conv1.bn if net.conv1 is not None else None
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d) and m is not root_bn:
            m.eval()
            m.train = lambda mode=True, _m=m: _m
            for p_ in m.parameters():
                p_.requires_grad_(False)


class ConvBN(nn.Module):
    """slim.conv2d under resnet_arg_scope: conv (no bias) + batch_norm [+ relu]; `same=True` is
    resnet_utils.conv2d_same."""

    def __init__(self, cin, cout, k, stride=1, relu=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
        self.bn = SlimBatchNorm2d(cout, eps=BN_EPS, momentum=1.0 - BN_DECAY)
        self.relu = relu

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x, inplace=True) if self.relu else x


class Bottleneck(nn.Module):
    """resnet_v1.bottleneck(inputs, depth, depth_bottleneck, stride)."""

    def __init__(self, depth_in, depth, depth_bottleneck, stride, final_relu: bool = True):
        super().__init__()
        self.stride = stride
        self.final_relu = final_relu     # False: return the residual sum (the consumer applies the ReLU)
        self.shortcut = None if depth == depth_in else ConvBN(depth_in, depth, 1, stride, relu=False)
        self.conv1 = ConvBN(depth_in, depth_bottleneck, 1)
        self.conv2 = ConvBN(depth_bottleneck, depth_bottleneck, 3, stride)
        self.conv3 = ConvBN(depth_bottleneck, depth, 1, relu=False)

    def forward(self, x):
        sc = subsample(x, self.stride) if self.shortcut is None else self.shortcut(x)
        y = sc + self.conv3(self.conv2(self.conv1(x)))
        return F.relu(y, inplace=True) if self.final_relu else y


BLOCKS = {   # resnet_v1.py:224-313: (depth, depth_bottleneck, units); stride 2 in the last unit
    'resnet_v1_50': [(256, 64, 3), (512, 128, 4), (1024, 256, 6), (2048, 512, 3)],
    'resnet_v1_101': [(256, 64, 3), (512, 128, 4), (1024, 256, 23), (2048, 512, 3)],
    'resnet_v1_152': [(256, 64, 3), (512, 128, 8), (1024, 256, 36), (2048, 512, 3)],
}


class ResNetV1(nn.Module):
    """`images [N,h,w,3]` (NHWC float, mean-subtracted like vgg_preprocessing.py:355-372) ->
    block4 feature map `[N,H,W,2048]` (NHWC view of a channels-last tensor, post-ReLU: the tap
    `resnet_v1_101/block4` of nets_factory.py:63-67)."""

    def __init__(self, name: str = 'resnet_v1_101', blocks: Sequence[Tuple[int, int, int]] = None,
                 include_root_block: bool = True, in_channels: int = 3, final_relu: bool = True):
        """`final_relu=False`: the very last unit returns its residual sum without the ReLU -- for a
        consumer that applies it on the fly (APA_FLAG_RELU_INPUT of the pooling op)."""
        super().__init__()
        self.name = name
        spec = list(blocks) if blocks is not None else BLOCKS[name]
        self.conv1 = ConvBN(in_channels, 64, 7, stride=2) if include_root_block else None
        depth_in = 64 if include_root_block else in_channels
        self.blocks = nn.ModuleList()
        for bi, (depth, neck, units) in enumerate(spec):
            last_block = bi == len(spec) - 1
            layers: List[nn.Module] = []
            for u in range(units):
                stride = 2 if (u == units - 1 and not last_block) else 1
                is_last = last_block and u == units - 1
                layers.append(Bottleneck(depth_in, depth, neck, stride, final_relu=final_relu or not is_last))
                depth_in = depth
            self.blocks.append(nn.Sequential(*layers))
        self.out_channels = depth_in
        self.to(memory_format=torch.channels_last)

    def forward(self, images: torch.Tensor, end_points: Dict[str, torch.Tensor] = None) -> torch.Tensor:
        x = images.permute(0, 3, 1, 2)                 # NHWC storage == channels-last NCHW view
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if self.conv1 is not None:
            x = max_pool_same(self.conv1(x), 3, 2)
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if end_points is not None:
                end_points['%s/block%d' % (self.name, i + 1)] = x.permute(0, 2, 3, 1)
        x = x.contiguous(memory_format=torch.channels_last)
        return x.permute(0, 2, 3, 1)                   # [N,H,W,C], contiguous: no copy for the op

    # ------------------------------------------------------------------------------------------
    # TF-slim variable names (warm start from the reference's checkpoints, README.md:41-43)
    # ------------------------------------------------------------------------------------------
    def tf_variable_map(self) -> Dict[str, Tuple[nn.Module, str]]:
        """TF variable name -> (module, attribute).  Conv weights are HWIO in TF (OIHW here)."""
        out: Dict[str, Tuple[nn.Module, str]] = {}

        def add(scope, cb: ConvBN):
            out[scope + '/weights'] = (cb.conv, 'weight')
            out[scope + '/BatchNorm/gamma'] = (cb.bn, 'weight')
            out[scope + '/BatchNorm/beta'] = (cb.bn, 'bias')
            out[scope + '/BatchNorm/moving_mean'] = (cb.bn, 'running_mean')
            out[scope + '/BatchNorm/moving_variance'] = (cb.bn, 'running_var')
        if self.conv1 is not None:
            add(self.name + '/conv1', self.conv1)
        for bi, blk in enumerate(self.blocks):
            for ui, unit in enumerate(blk):
                pre = '%s/block%d/unit_%d/bottleneck_v1' % (self.name, bi + 1, ui + 1)
                if unit.shortcut is not None:
                    add(pre + '/shortcut', unit.shortcut)
                add(pre + '/conv1', unit.conv1)
                add(pre + '/conv2', unit.conv2)
                add(pre + '/conv3', unit.conv3)
        return out

    @torch.no_grad()
    def load_tf_variables(self, variables: Dict[str, np.ndarray], strict: bool = True) -> List[str]:
        """Copy numpy arrays keyed by TF variable name into the module; returns the names used."""
        used = []
        for name, (mod, attr) in self.tf_variable_map().items():
            if name not in variables:
                if strict:
                    raise KeyError('missing TF variable ' + name)
                continue
            v = torch.as_tensor(np.asarray(variables[name]))
            if attr == 'weight' and isinstance(mod, nn.Conv2d):
                v = v.permute(3, 2, 0, 1)              # HWIO -> OIHW
            getattr(mod, attr).copy_(v.to(getattr(mod, attr).dtype))
            used.append(name)
        return used
