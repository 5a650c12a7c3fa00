#!/usr/bin/env python
"""Golden vectors for the BACKBONE boundary: the conv5 tap
`end_points['resnet_v1_<n>/block4']` that the head reads (nets_factory.py:63-67,136-140), produced by the
REFERENCE'S OWN graph code.

Executed from /root/reference behind tests/golden/tf1_shim.py (float64):
  models/slim/nets/resnet_v1.py     -- bottleneck, resnet_v1, resnet_v1_50 / _101 (block tables, where the stride sits)
  models/slim/nets/resnet_utils.py  -- Block, subsample, conv2d_same, stack_blocks_dense, resnet_arg_scope
The stand-in supplies what those files call in TensorFlow / slim: Conv2D with 'SAME' / 'VALID' padding and
dilation, max pooling, batch normalisation in inference (moving statistics) and training (batch moments) mode,
tf.pad, variable scopes and the named-output collections the end points come from.

Variables are NOT stored (25 M / 44 M values): every variable `v` of case `c` is
    f32( RandomState(crc32(c | tf name of v)).randn(*shape) * sigma ),   sigma = sqrt(2 / fan_in) for conv weights,
    gamma = 1 + 0.1 randn, beta = 0.1 randn, moving_mean = 0.1 randn, moving_variance = 1 + 0.1 |randn|
(`variable_value`, imported by the test).  Stored: the input images, the block4 map, and for every other end
point its shape, sum and sum of squares.

Run in the build container:   python tests/golden/make_backbone_reference.py
Output: tests/golden/ref_backbone.npz.  Test infrastructure only.
"""
from __future__ import annotations

import json
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf1_shim as tfs                                   # noqa: E402
import make_head_reference as mhr                        # noqa: E402

REF = mhr.REF

CASES = [
    # odd, non-square images: 'SAME' padding is asymmetric on some layers and not on others
    dict(name='resnet_v1_50_eval', model='resnet_v1_50', shape=(2, 65, 97, 3), is_training=False),
    dict(name='resnet_v1_50_even_eval', model='resnet_v1_50', shape=(1, 64, 96, 3), is_training=False),
    dict(name='resnet_v1_101_eval', model='resnet_v1_101', shape=(1, 75, 59, 3), is_training=False),
    dict(name='resnet_v1_50_train', model='resnet_v1_50', shape=(3, 49, 65, 3), is_training=True),
    # cfg.NET.TRAIN_TOP_BN (resnet_v1.py:191-203): only the root block's batch norm is in training mode
    dict(name='resnet_v1_101_train_top_bn', model='resnet_v1_101', shape=(2, 33, 49, 3), is_training=True,
         train_top_bn=True),
]


from backbone_values import variable_value          # noqa: E402  (the documented formula, shared with the test)


def load_backbone_reference():
    cfgmod, nf, _loss = mhr.load_reference()
    # the real backbone file instead of the head generator's stub module
    rv1 = mhr._exec_ref(os.path.join(mhr.NETS, 'resnet_v1.py'), 'nets.resnet_v1')
    sys.modules['nets'].resnet_v1 = rv1
    return cfgmod, nf, rv1


def run_case(rv1, nf, c):
    name = c['name']
    g = tfs.Graph(lambda vn, shape, desc: variable_value(name, vn, shape), None)
    tfs.set_graph(g)
    r = np.random.RandomState(zlib.crc32(('%s|images' % name).encode()) & 0x7fffffff)
    images = (r.randn(*c['shape']) * 50.0).astype(np.float32)          # mean-subtracted pixels
    fn = getattr(rv1, c['model'])
    # the call of nets_factory.network_fn (:126-131): func(images, num_classes, is_training=..., train_top_bn=...)
    with tfs.arg_scope(rv1.resnet_arg_scope()):
        kw = {'train_top_bn': c.get('train_top_bn', False)} if c['model'] == 'resnet_v1_101' else {}
        net, end_points = fn(tfs.Tensor(torch.from_numpy(images.astype(np.float64))), 11,
                             is_training=c['is_training'], **kw)
    tap = c['model'] + '/block4'
    assert nf.last_conv_map['resnet_v1_101'] == 'resnet_v1_101/block4'      # the head's tap (nets_factory.py:63-67)
    out = {'in/images': images, 'out/block4': end_points[tap].v.detach().numpy().astype(np.float32)}
    stats = {}
    for k, t in end_points.items():
        if isinstance(t, tfs.Tensor):
            v = t.v.detach().numpy()
            stats[k] = dict(shape=list(v.shape), sum=float(v.sum()), sumsq=float((v * v).sum()))
    updates = {}
    for kind, val in g.get_collection(tfs.GraphKeys.UPDATE_OPS):
        updates.setdefault(kind, []).append(val.numpy())
    if updates:          # batch-norm moving statistics after one training step, first and last layer
        out['out/update/moving_mean/first'] = updates['moving_mean'][0]
        out['out/update/moving_mean/last'] = updates['moving_mean'][-1]
        out['out/update/moving_variance/first'] = updates['moving_variance'][0]
        out['out/update/moving_variance/last'] = updates['moving_variance'][-1]
    meta = dict(case=name, model=c['model'], is_training=c['is_training'], train_top_bn=bool(c.get('train_top_bn', False)), tap=tap, end_points=stats,
                var_order=g.var_order, var_shapes={vn: list(g.variables[vn].shape) for vn in g.var_order},
                n_updates={k: len(v) for k, v in updates.items()},
                logits_shape=list(net.v.shape))
    out['meta'] = np.array(json.dumps(meta, sort_keys=True))
    tfs.set_graph(None)
    return out


# ------------------------------------------------------------------------------------------------------------
# BASELINE configs[0] end to end: experiments/001_MPII_ResNet.yaml through the reference's get_network_fn with the
# REAL backbone (no stub), gen_losses and tf.gradients -- images in, logits / losses / gradients out.
# ------------------------------------------------------------------------------------------------------------
E2E_CASES = [
    dict(name='cfg001_train_e2e', yaml='001_MPII_ResNet.yaml', shape=(4, 49, 65, 3), K=7, is_training=True),
    dict(name='cfg001_eval_e2e', yaml='001_MPII_ResNet.yaml', shape=(2, 65, 49, 3), K=7, is_training=False),
    # NET.DROPOUT >= 0 IS forwarded to the backbone (nets_factory.py:127-129): keep 0.5 on the pooled vector
    dict(name='cfg001_train_dropout_e2e', yaml='001_MPII_ResNet.yaml', shape=(4, 49, 65, 3), K=7, is_training=True,
         net={'DROPOUT': 0.5}, libmask=(42, 0)),
    # BASELINE configs[1] / [2] at small size, images in: backbone -> conv5 tap -> attention head -> losses.  The head's
    # dropout (keep 0.2 on the [N,h,w,2048] map) runs on the library's own mask for (seed, offset).
    dict(name='cfg002_train_e2e', yaml='002_MPII_ResNet_withAttention.yaml', shape=(4, 49, 65, 3), K=7,
         is_training=True, libmask=(42, 0)),
    dict(name='cfg003_train_e2e', yaml='003_MPII_ResNet_withPoseAttention.yaml', shape=(4, 49, 65, 3), K=7,
         is_training=True, libmask=(43, 5)),
]
FULL_GRADS = ['PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/weights', 'PosePrelogitsBasedAttention/Conv2d_PrePose_Attn/biases',
              'PosePrelogitsBasedAttention/Conv/weights', 'PosePrelogitsBasedAttention/Conv/biases',
              'PoseLogits/Conv2d_1c_1x1/weights', 'PoseLogits/Conv2d_1c_1x1/biases', 'PoseLogits/ExtraConv2d_1x1/biases',
              'resnet_v1_101/logits/weights', 'resnet_v1_101/logits/biases', 'resnet_v1_101/conv1/BatchNorm/gamma',
              'resnet_v1_101/conv1/BatchNorm/beta', 'resnet_v1_101/block4/unit_3/bottleneck_v1/conv3/BatchNorm/gamma',
              'resnet_v1_101/block1/unit_1/bottleneck_v1/shortcut/BatchNorm/beta']


def run_e2e_case(cfgmod, nf, lossmod, rv1, defaults, c):
    import copy
    name = c['name']
    mhr.reset_cfg(cfgmod, defaults)
    cfg = cfgmod.cfg
    cfgmod.cfg_from_file(os.path.join(REF, 'experiments', c['yaml']))
    mhr.merge(cfg.NET, c.get('net', {}))
    K, J = c['K'], 16
    wd = float(cfg.TRAIN.WEIGHT_DECAY)
    count = [0]

    def uniform_fn(shape, what):
        # the LIBRARY'S OWN keep mask for (seed, offset) as tf.nn.dropout's uniforms (see make_head_reference.py,
        # `libmask`): the product then runs this case on its own counter hash, nothing replayed
        import apa_keep_mask
        assert what == 'dropout' and count[0] == 0
        count[0] += 1
        seed, offset = c['libmask']
        d = float(cfg.NET.DROPOUT)
        attention = bool(cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION)
        keep_prob = (0.2 if d < 0 else 1.0 - d) if attention else 1.0 - d      # head rule (:143-146) / backbone kwarg (:127-129)
        keep = apa_keep_mask.keep_mask(shape, keep_prob, seed, offset)
        return np.where(keep == 1, 0.9, 0.1)

    g = tfs.Graph(lambda vn, shape, desc: variable_value(name, vn, shape), uniform_fn)
    tfs.set_graph(g)
    nf.networks_map['resnet_v1_101'] = rv1.resnet_v1_101                 # the real backbone
    r = np.random.RandomState(zlib.crc32(('%s|images' % name).encode()) & 0x7fffffff)
    images_np = (r.randn(*c['shape']) * 50.0).astype(np.float32)
    labels = r.randint(0, K, size=(c['shape'][0],))
    images = tfs.Tensor(torch.from_numpy(images_np.astype(np.float64)).requires_grad_(True))
    network_fn = nf.get_network_fn(cfg.MODEL_NAME, K, J, cfg, weight_decay=wd, is_training=c['is_training'])
    logits, end_points = network_fn(images)
    use_pose = bool(cfg.TRAIN.LOSS_FN_POSE)
    lab_pose = valid = None
    if use_pose:
        pl = end_points['PoseLogits']
        lab_pose = mhr.f32(r.rand(*pl.v.shape))
        valid = r.rand(pl.v.shape[0], J) > 0.3
    lossmod.gen_losses(tfs.Tensor(torch.from_numpy(labels)), logits, cfg.TRAIN.LOSS_FN_ACTION, K,
                       cfg.TRAIN.LOSS_FN_ACTION_WT,
                       tfs.Tensor(torch.from_numpy(lab_pose)) if use_pose else None,
                       end_points['PoseLogits'] if use_pose else None, cfg.TRAIN.LOSS_FN_POSE if use_pose else '',
                       tfs.Tensor(torch.from_numpy(valid)) if use_pose else None,
                       cfg.TRAIN.LOSS_FN_POSE_WT, end_points, cfg)
    losses = g.get_collection(tfs.GraphKeys.LOSSES)
    regs = g.get_collection(tfs.GraphKeys.REGULARIZATION_LOSSES)
    total = sum(l.v for l in losses) + sum(l.v for l in regs)
    total.backward()
    weights = [vn for vn in g.var_order if vn.endswith('/weights')]
    assert len(regs) == len(weights)                                      # one L2 term per conv `weights`
    reg_groups = {'backbone': 0.0, 'logits': 0.0, 'PoseLogits': 0.0, 'attention': 0.0}
    for vn, l in zip(weights, regs):
        key = 'PoseLogits' if vn.startswith('PoseLogits/') else ('attention' if vn.startswith('PosePrelogits') else (
            'logits' if '/logits/' in vn else 'backbone'))
        reg_groups[key] += float(l.v.detach())
    out = {'in/images': images_np, 'in/labels_action': labels.astype(np.int64),
           'out/logits': logits.v.detach().numpy(), 'out/losses': np.array([float(l.v.detach()) for l in losses]),
           'out/total': np.float64(float(total.detach())), 'grad/images': images.v.grad.numpy().astype(np.float32),
           'out/block4': end_points[cfg.MODEL_NAME + '/block4'].v.detach().numpy().astype(np.float32)}
    if use_pose:
        out['in/labels_pose'] = lab_pose.astype(np.float32)
        out['in/labels_pose_valid'] = valid
    grad_stats = {}
    for vn in g.var_order:
        v = g.variables[vn]
        if v.requires_grad:
            gr = np.zeros(v.shape) if v.grad is None else v.grad.numpy()
            data = gr - (wd * v.detach().numpy() if vn.endswith('/weights') else 0.0)      # without the L2 term
            grad_stats[vn] = dict(sum=float(data.sum()), sumsq=float((data * data).sum()), none=v.grad is None)
            if vn in FULL_GRADS:
                out['grad/var/' + vn] = data.astype(np.float32) if data.size > 4096 else data   # big ones at float32
    draws = []
    for i, d in enumerate(g.random_draws):
        keep = np.floor(d['keep_prob'] + d['uniform']).astype(np.uint8)
        out['rand/%d/keep_bits' % i] = np.packbits(keep.reshape(-1))
        draws.append({'kind': d['kind'], 'keep_prob': d['keep_prob'], 'shape': list(keep.shape)})
    updates = {}
    for kind, val in g.get_collection(tfs.GraphKeys.UPDATE_OPS):
        updates.setdefault(kind, []).append(val.numpy())
    if updates:
        out['out/update/moving_mean/first'] = updates['moving_mean'][0]
        out['out/update/moving_variance/first'] = updates['moving_variance'][0]
        out['out/update/moving_mean/last'] = updates['moving_mean'][-1]
        out['out/update/moving_variance/last'] = updates['moving_variance'][-1]
    meta = dict(case=name, model=cfg.MODEL_NAME, num_classes=K, is_training=c['is_training'], weight_decay=wd,
                dropout=float(cfg.NET.DROPOUT), attention=bool(cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION),
                net={k: v for k, v in cfg.NET.items() if not isinstance(v, dict)}, libmask=list(c['libmask']) if c.get('libmask') else None, reg_groups=reg_groups, grad_stats=grad_stats, draws=draws,
                var_order=g.var_order, var_shapes={vn: list(g.variables[vn].shape) for vn in g.var_order},
                end_points=sorted(k for k, t in end_points.items() if isinstance(t, tfs.Tensor)),
                n_losses=len(losses), n_updates={k: len(v) for k, v in updates.items()},
                train_cfg={k: cfg.TRAIN[k] for k in ('LOSS_FN_POSE', 'LOSS_FN_POSE_WT', 'LOSS_FN_POSE_SAMPLED',
                                                     'LOSS_FN_ACTION', 'LOSS_FN_ACTION_WT', 'WEIGHT_DECAY')})
    out['meta'] = np.frombuffer(json.dumps(meta, sort_keys=True, default=str).encode(), dtype=np.uint8)   # utf-8 bytes
    tfs.set_graph(None)
    return out


def e2e_meta(blobs, name):
    return json.loads(bytes(blobs[name + '/meta']).decode())


def generate_e2e(names=None):
    import copy
    cfgmod, nf, rv1 = load_backbone_reference()
    lossmod = sys.modules['refloss']
    defaults = copy.deepcopy(cfgmod.cfg)
    blobs = {}
    for c in E2E_CASES:
        if names is None or c['name'] in names:
            for k, v in run_e2e_case(cfgmod, nf, lossmod, rv1, defaults, c).items():
                blobs['%s/%s' % (c['name'], k)] = v
    blobs['cases'] = np.array(json.dumps([c['name'] for c in E2E_CASES if names is None or c['name'] in names]))
    mhr.reset_cfg(cfgmod, defaults)
    return blobs


def generate(names=None):
    _cfgmod, nf, rv1 = load_backbone_reference()
    blobs = {}
    for c in CASES:
        if names is None or c['name'] in names:
            for k, v in run_case(rv1, nf, c).items():
                blobs['%s/%s' % (c['name'], k)] = v
    blobs['cases'] = np.array(json.dumps([c['name'] for c in CASES if names is None or c['name'] in names]))
    return blobs


if __name__ == '__main__':
    blobs = generate()
    path = os.path.join(HERE, 'ref_backbone.npz')
    np.savez_compressed(path, **blobs)
    for c in CASES:
        m = json.loads(str(blobs[c['name'] + '/meta']))
        print('%-26s block4 %s  %d variables  %d end points' % (
            c['name'], m['end_points'][m['tap']]['shape'], len(m['var_order']), len(m['end_points'])))
    print('wrote', path, os.path.getsize(path), 'bytes')
    blobs = generate_e2e()
    path = os.path.join(HERE, 'ref_e2e.npz')
    np.savez_compressed(path, **blobs)
    for c in E2E_CASES:
        m = e2e_meta(blobs, c['name'])
        print('%-26s logits %s  losses %s  reg %s  draws %d' % (
            c['name'], list(blobs[c['name'] + '/out/logits'].shape), blobs[c['name'] + '/out/losses'].round(4).tolist(),
            {k: round(v, 4) for k, v in m['reg_groups'].items()}, len(m['draws'])))
    print('wrote', path, os.path.getsize(path), 'bytes')
