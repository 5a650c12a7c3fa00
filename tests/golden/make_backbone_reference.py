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
