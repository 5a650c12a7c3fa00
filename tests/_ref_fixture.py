"""Reader of the reference-generated fixtures tests/golden/ref_head_*.npz / ref_losses.npz (written by
tests/golden/make_head_reference.py, which EXECUTES /root/reference's nets_factory.py and loss.py) and the
glue that replays one fixture through the CPU oracle.  Test infrastructure."""
import glob
import json
import os

import numpy as np
import torch

from oracle import attn_pool_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
import sys
if GOLD not in sys.path:
    sys.path.insert(0, GOLD)
import apa_digest          # noqa: E402  (tests/golden: digest / sample / bf16 rounding shared with the generator)
import apa_keep_mask       # noqa: E402  (numpy twin of the library's dropout stream)
PRE = 'PosePrelogitsBasedAttention/'
P = 'USE_POSE_PRELOGITS_BASED_ATTENTION'


def head_fixture_paths():
    return sorted(glob.glob(os.path.join(GOLD, 'ref_head_*.npz')))


def big_fixture_paths():
    """the benchmark-shape cases (make_head_reference.BIG_CASES): inputs by seed, large outputs by digest"""
    return sorted(glob.glob(os.path.join(GOLD, 'refbig_*.npz')))


def case_id(path):
    b = os.path.basename(path)
    return b[len('refbig_'):-4] if b.startswith('refbig_') else b[len('ref_head_'):-4]


class HeadFixture(object):
    def __init__(self, path=None, arrays=None, meta=None):
        if path is not None:
            d = np.load(path)
            arrays = {k: d[k] for k in d.files if k != 'meta'}
            meta = json.loads(str(d['meta']))
        self.arrays = arrays
        self.meta = meta
        self.name = self.meta['case']
        self.f32_keys = set(self.meta['f32_keys'])
        self.quant = self.meta.get('quant')
        rnd = apa_digest.bf16_round if self.quant == 'bf16' else (lambda a: a)
        if 'inseed/images' in self.arrays:
            # X = round(relu(RandomState(seed).randn(*shape))) -- the generator's first draw from its input stream
            spec = [int(v) for v in self.arrays['inseed/images']]
            x = np.random.RandomState(spec[0]).randn(*spec[3:])
            if spec[1]:
                x = np.maximum(x, 0)
            x = apa_digest.bf16_round(x) if spec[2] else x.astype(np.float32).astype(np.float64)
            if 'inpatch/images_idx' in self.arrays:       # make_head_reference.gate_safe_inputs: a few features moved
                x = np.array(x, dtype=np.float64)
                x.reshape(-1)[self.arrays['inpatch/images_idx']] = self.arrays['inpatch/images_val'].astype(np.float64)
            chk = self.arrays['insum/images']
            assert abs(x.sum() - chk[0]) <= 1e-9 * abs(chk[0]) and abs((x ** 2).sum() - chk[1]) <= 1e-12 * chk[1]
            self.arrays['in/images'] = x.astype(np.float32)
        self.variables = {}
        for vn in self.meta['var_order']:
            if 'var/' + vn in self.arrays:
                self.variables[vn] = self.arrays['var/' + vn].astype(np.float64)
            else:
                # a regulariser-only variable stored as the seed of the generator's documented formula
                # (make_head_reference.make_value_fn, 'trained' mode): f32(RandomState(seed).randn(*shape)/sqrt(fan_in))
                spec = self.arrays['varseed/' + vn]
                seed, fan_in, shape = int(spec[0]), int(spec[1]), [int(s) for s in spec[2:]]
                v = rnd((np.random.RandomState(seed).randn(*shape) / np.sqrt(fan_in)).astype(np.float32).astype(np.float64))
                chk = self.arrays['varsum/' + vn]
                assert abs(v.sum() - chk[0]) <= 1e-9 * max(1.0, abs(chk[0])) and \
                    abs((v ** 2).sum() - chk[1]) <= 1e-12 * chk[1], 'regenerated %s does not match its checksum' % vn
                self.variables[vn] = v

    # ---- inputs
    @property
    def net(self):
        return self.meta['net']

    def flag(self, suffix=''):
        return self.net[P + suffix]

    @property
    def images(self):
        return self.arrays['in/images'].astype(np.float64)

    @property
    def pose_tap(self):
        a = self.arrays.get('in/pose_tap')
        return None if a is None else a.astype(np.float64)

    @property
    def keep_prob(self):
        d = float(self.net['DROPOUT'])
        return 0.2 if d < 0 else 1.0 - d               # nets_factory.py:143-146

    def dropout_mask(self):
        """the {0,1} keep mask tf.nn.dropout drew (floor(keep + U)), or None in evaluation mode"""
        for i, dr in enumerate(self.meta['draws']):
            if dr['kind'] == 'dropout' and 'rand/%d/libmask' % i in self.arrays:
                seed, offset, count = (int(v) for v in self.arrays['rand/%d/libmask' % i])
                keep = apa_keep_mask.keep_mask(tuple(dr['shape']), dr['keep_prob'], seed, offset)
                assert int(keep.sum()) == count
                return keep
            if dr['kind'] == 'dropout':
                n = int(np.prod(dr['shape']))
                bits = np.unpackbits(self.arrays['rand/%d/keep_bits' % i])[:n]
                assert abs(dr['keep_prob'] - self.keep_prob) < 1e-12
                return bits.reshape(dr['shape'])
        return None

    def var(self, name, squeeze_hw=True):
        v = self.variables[name]
        return v.reshape(v.shape[-2:]) if (squeeze_hw and v.ndim == 4) else v

    def expected(self, key):
        return self.arrays[key]

    def has(self, key):
        return key in self.arrays or 'digest/' + key in self.arrays

    def output_keys(self):
        """every stored output / gradient, whether in full or as digest + sample"""
        ks = [k for k in self.arrays if k.startswith('out/') or k.startswith('grad/')]
        ks += [k[len('digest/'):] for k in self.arrays if k.startswith('digest/')]
        return ks

    def check(self, key, got, tol, what='', floor=1e-30, tol_proj=None):
        """`got` against the stored value of `key`: in full (max abs error <= tol * max|expected|), or -- for the
        tensors a benchmark-shape fixture keeps as digest + sample -- the 4096 sampled values at `tol` and the
        two whole-tensor projections at `tol_proj` (default 4 x tol) of the tensor's l2 norm."""
        got = np.asarray(got, dtype=np.float64)
        if 'digest/' + key in self.arrays:
            apa_digest.check(got, self.arrays['digest/' + key], self.arrays['sample/' + key], tol,
                             4 * tol if tol_proj is None else tol_proj, what or key)
            return
        exp = np.asarray(self.arrays[key], dtype=np.float64)
        err = float(np.abs(got.reshape(exp.shape) - exp).max()) if exp.size else 0.0
        scale = max(float(np.abs(exp).max()) if exp.size else 0.0, floor)
        assert err <= tol * scale, '%s: max abs err %.3e > %.1e * %.3e' % (what or key, err, tol, scale)

    def tol(self, key, tight=1e-12):
        """float64-stored tensors are compared at `tight`; the ones stored as float32 at storage rounding"""
        return 1.5e-7 if key in self.f32_keys else tight


def run_oracle(fx: HeadFixture, dtype=torch.float64):
    """Replays fixture `fx` through oracle/attn_pool_oracle.py.  Returns a dict with the fixture's own keys
    ('out/logits', 'out/ep/<name>', 'out/losses', 'out/reg_losses', 'out/total', 'grad/images',
    'grad/pose_tap', 'grad/var/<tf name>', 'out/update/moving_mean|variance')."""
    m = fx.meta
    K, J = m['num_classes'], m['num_pose_keypoints']
    train = m['is_training']
    model = m['model']
    arg_scope = {'resnet_v1_101': 'resnet', 'vgg_16': 'vgg', 'inception_v2_tsn': 'inception_v2_tsn',
                 'inception_v3': 'inception_v3'}[model]
    leaf = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).to(dtype).requires_grad_(True)
    V = {vn: leaf(fx.var(vn)) for vn in m['var_order']}
    images = leaf(fx.images)
    x = images
    frames = 1
    if x.dim() == 5:                                                   # nets_factory.py:121-125
        frames = x.shape[1]
        x = x.reshape(-1, *x.shape[2:])
    pose_tap = leaf(fx.pose_tap) if fx.pose_tap is not None else None
    flags = orc.AttnFlags(single_layer_att=bool(fx.flag('_SINGLE_LAYER_ATT')), softmax_att=bool(fx.flag('_SOFTMAX_ATT')),
                          relu_att=bool(fx.flag('_RELU_ATT')), per_class=bool(fx.flag('_PER_CLASS')),
                          rank=int(fx.flag('_RANK')), with_pose_feat=bool(fx.flag('_WITH_POSE_FEAT')),
                          with_pose_feat_2layer=bool(fx.flag('_WITH_POSE_FEAT_2LAYER')))
    pre, pl = orc.pose_logits_head(x if pose_tap is None else pose_tap,
                                   V['PoseLogits/ExtraConv2d_1x1/weights'], V['PoseLogits/ExtraConv2d_1x1/biases'],
                                   V['PoseLogits/Conv2d_1c_1x1/weights'], V['PoseLogits/Conv2d_1c_1x1/biases'])
    R = flags.rank
    att_names = ['Conv2d_PrePose_Attn' + (str(r) if r else '') for r in range(R)]
    two = flags.with_pose_feat and flags.with_pose_feat_2layer
    # unnamed convs take 'Conv', 'Conv_1', ... in creation order: the _2LAYER conv comes first (:291-294)
    td_names = ['Conv' + ('_%d' % (r + two) if (r + two) else '') for r in range(R)]
    kw = {}
    if two:
        kw['pose_feat_w'] = V[PRE + 'Conv/weights']
        kw['arg_scope'] = arg_scope
        if arg_scope in ('resnet', 'inception_v3'):
            kw['pose_feat_bn'] = (V.get(PRE + 'Conv/BatchNorm/gamma'), V[PRE + 'Conv/BatchNorm/beta'])
        else:
            kw['pose_feat_b'] = V[PRE + 'Conv/biases']
    mask = fx.dropout_mask()
    logits, ep = orc.attentional_pooling(
        x, pre, pl, [V[PRE + n + '/weights'] for n in att_names], [V[PRE + n + '/biases'] for n in att_names],
        [V[PRE + n + '/weights'] for n in td_names], [V[PRE + n + '/biases'] for n in td_names], flags,
        is_training=train, keep_prob=fx.keep_prob, dropout_mask=None if mask is None else torch.from_numpy(mask), **kw)
    ep['PoseLogits'] = pl
    if frames > 1:                                                     # :354-374
        tw = V.get('TemporalAttention/Conv/weights')
        logits, ep2 = orc.frame_pooling(logits, frames, tw, V.get('TemporalAttention/Conv/biases'))
        ep.update(ep2)

    class _Cfg(object):
        class TRAIN(object):
            LOSS_FN_POSE_SAMPLED = bool(m['train_cfg']['LOSS_FN_POSE_SAMPLED'])
    tc = m['train_cfg']
    use_pose = bool(tc['LOSS_FN_POSE'])
    losses = orc.gen_losses(
        torch.from_numpy(fx.arrays['in/labels_action']), logits, tc['LOSS_FN_ACTION'], K, tc['LOSS_FN_ACTION_WT'],
        torch.from_numpy(fx.arrays['in/labels_pose'].astype(np.float64)).to(dtype) if use_pose else None,
        pl if use_pose else None, tc['LOSS_FN_POSE'] if use_pose else '',
        torch.from_numpy(fx.arrays['in/labels_pose_valid']) if use_pose else None, tc['LOSS_FN_POSE_WT'], ep, _Cfg)
    # slim.l2_regularizer on every conv `weights` variable, in creation order (REGULARIZATION_LOSSES)
    regs = [orc.l2_regularizer([V[vn]], m['weight_decay']) for vn in m['var_order'] if vn.endswith('/weights')] \
        if m['weight_decay'] > 0 else []
    total = sum(losses) + (sum(regs) if regs else 0.0)
    total.backward()
    out = {'out/logits': logits, 'out/losses': torch.stack([l.reshape(()) for l in losses]),
           'out/reg_losses': torch.stack(regs) if regs else torch.zeros(0), 'out/total': total,
           'grad/images': images.grad}
    if pose_tap is not None:
        out['grad/pose_tap'] = pose_tap.grad
    for k, v in ep.items():
        out['out/ep/' + k] = v
    for vn in m['trainable']:
        g = V[vn].grad
        out['grad/var/' + vn] = (torch.zeros_like(V[vn]) if g is None else g).reshape(fx.variables[vn].shape)
    if two and arg_scope in ('resnet', 'inception_v3'):                # UPDATE_OPS of the batch-norm
        decay = 0.997 if arg_scope == 'resnet' else 0.9997             # resnet_utils.py:210 / inception_utils.py:34
        y = orc.conv1x1(pl.detach(), V[PRE + 'Conv/weights'].detach(), None)
        mean, var = y.mean(dim=(0, 1, 2)), y.var(dim=(0, 1, 2), unbiased=False)
        mm, mv = V[PRE + 'Conv/BatchNorm/moving_mean'].detach(), V[PRE + 'Conv/BatchNorm/moving_variance'].detach()
        out['out/update/moving_mean'] = (mm - (1 - decay) * (mm - mean))[None]
        out['out/update/moving_variance'] = (mv - (1 - decay) * (mv - var))[None]
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}


def train_fixture_paths():
    return sorted(glob.glob(os.path.join(GOLD, 'ref_train_*.npz')))


class TrainFixture(object):
    """tests/golden/ref_train_<case>.npz (make_train_reference.py: the reference's model_deploy.py + the training
    pieces of src/train.py): initial variables, the batches in dequeue order, per session.run the summed clone
    gradients, per parameter update the learning rate / loss / variables."""

    def __init__(self, path):
        d = np.load(path)
        self.arrays = {k: d[k] for k in d.files if k != 'meta'}
        self.meta = json.loads(str(d['meta']))
        self.name = self.meta['case']
        self.f32_keys = set(self.meta['f32_keys'])

    def initial_variables(self):
        return {vn: self.arrays['var0/' + vn].astype(np.float64) for vn in self.meta['var_order']}

    def tol(self, key, tight=1e-11):
        return 2e-7 if key in self.f32_keys else tight

    def clone_fixture(self, variables, batch, draw, weight_decay=0.0):
        """The head fixture of ONE clone in ONE run: `variables` {tf name: value} are the current weights, `batch`
        / `draw` index this file's batches / dropout draws.  clone_fn (src/train.py:395-398) concatenates the
        label tensors over the batch axis."""
        m = self.meta
        a = {'in/images': self.arrays['batch/%d/images' % batch],
             'in/labels_action': self.arrays['batch/%d/labels_action' % batch]}
        lp = self.arrays['batch/%d/labels_pose' % batch]
        a['in/labels_pose'] = lp.reshape((-1,) + lp.shape[2:])
        lv = self.arrays['batch/%d/labels_pose_valid' % batch]
        a['in/labels_pose_valid'] = lv.reshape((-1,) + lv.shape[2:])
        for vn, v in variables.items():
            a['var/' + vn] = np.asarray(v, dtype=np.float64)
        a['rand/0/keep_bits'] = self.arrays['rand/%d/keep_bits' % draw]
        meta = dict(case='%s/batch%d' % (self.name, batch), f32_keys=[], var_order=m['var_order'],
                    trainable=m['var_order'], num_classes=m['num_classes'],
                    num_pose_keypoints=m['num_pose_keypoints'], is_training=True, model=m['model'], net=m['net'],
                    train_cfg=m['train_cfg'], weight_decay=weight_decay, draws=[m['draws'][draw]], reg_only_grad=[])
        return HeadFixture(arrays=a, meta=meta)


def _cfg_for(fx):
    from attentionalpoolingaction_amd import config as apa_config
    apa_config.reset_cfg()
    net = {k: v for k, v in fx.meta['net'].items() if k != 'USE_POSE_ATTENTION_LOGITS_DIMS'}
    return apa_config.cfg_from_dict({'MODEL_NAME': fx.meta['model'], 'NET': net, 'TRAIN': dict(fx.meta['train_cfg'])})


def build_head(fx, device='cpu', **kw):
    """the product module for a fixture's configuration (no kernels run: construction only on CPU)"""
    from attentionalpoolingaction_amd import nets_factory
    cfg = _cfg_for(fx)
    shape = fx.arrays['in/images'].shape
    kw.setdefault('in_channels', shape[-1])
    if fx.pose_tap is not None:
        kw.setdefault('pose_in_channels', fx.pose_tap.shape[-1])
    return nets_factory.get_network_fn(fx.meta['model'], fx.meta['num_classes'], fx.meta['num_pose_keypoints'], cfg,
                                       weight_decay=fx.meta['weight_decay'], is_training=fx.meta['is_training'],
                                       device=device, **kw), cfg


def module_tf_names(network_fn):
    """{tf variable name: tensor} of the head (+ the TemporalAttention conv) as the product declares them"""
    head = network_fn.head
    names = head.tf_variable_names()
    sd = dict(head.named_parameters())
    sd.update(dict(head.named_buffers()))
    out = {tfn: sd[attr] for attr, tfn in names.items() if attr in sd}
    if network_fn.temporal is not None:
        out['TemporalAttention/Conv/weights'] = network_fn.temporal['weights']
        out['TemporalAttention/Conv/biases'] = network_fn.temporal['biases']
    return out


def load_loss_cases():
    d = np.load(os.path.join(GOLD, 'ref_losses.npz'))
    names = json.loads(str(d['cases']))
    out = []
    for n in names:
        c = {k[len(n) + 1:]: d[k] for k in d.files if k.startswith(n + '/')}
        c['meta'] = json.loads(str(c['meta']))
        c['name'] = n
        out.append(c)
    return out
