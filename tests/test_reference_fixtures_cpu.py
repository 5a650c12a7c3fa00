"""The oracle against fixtures produced by the REFERENCE'S OWN head / loss code.

tests/golden/ref_head_*.npz and ref_losses.npz are written by tests/golden/make_head_reference.py, which
exec's /root/reference/models/slim/nets/nets_factory.py (network_fn), src/loss.py (gen_losses), src/config.py
and the backbones' arg_scope functions behind a float64 TensorFlow/slim stand-in.  These tests pin
oracle/attn_pool_oracle.py -- the restatement every HIP parity test is measured against -- to those fixtures
at float64 round-off, and pin the product's variable-name / initialiser / regulariser tables to what the
reference's graph construction actually created."""
import json
import os

import numpy as np
import pytest
import torch

import _ref_fixture as rf
from oracle import attn_pool_oracle as orc

HEAD_PATHS = rf.head_fixture_paths()


def _close(got, exp, tol, what):
    got = np.asarray(got, dtype=np.float64).reshape(np.asarray(exp).shape)
    exp = np.asarray(exp, dtype=np.float64)
    # gradients that are exactly zero in exact arithmetic (the bias of a spatial softmax) are ~1e-17 noise on
    # both sides: the scale floor keeps the comparison meaningful
    scale = max(float(np.abs(exp).max()) if exp.size else 0.0, 1e-3)
    err = float(np.abs(got - exp).max()) if exp.size else 0.0
    assert err <= tol * scale, '%s: max abs err %.3e > %.1e * %.3e' % (what, err, tol, scale)


def test_there_are_reference_fixtures_for_every_flag_family():
    names = {rf.case_id(p) for p in HEAD_PATHS}
    for need in ('cfg002_eval', 'cfg002_train', 'cfg003_train', 'softmax_train', 'relu_train', 'perclass_train',
                 'rank2_train', 'rank3_relu_train', 'posefeat_train', 'posefeat_2layer_train',
                 'video_temporal_att_train', 'vgg16_train', 'tsn_separate_pose_tap_train'):
        assert need in names, need


@pytest.mark.parametrize('path', HEAD_PATHS, ids=rf.case_id)
def test_oracle_matches_reference_head_fixture(path):
    """logits, every end point, every loss term, the regularisation terms, the total loss and ALL gradients
    (feature map, pose tap, every trainable variable) of the reference graph == the oracle, to 1e-12 (tensors
    the fixture stores as float32: to float32 storage rounding)."""
    fx = rf.HeadFixture(path)
    got = rf.run_oracle(fx)
    checked = 0
    for key, exp in fx.arrays.items():
        if not (key.startswith('out/') or key.startswith('grad/')):
            continue
        assert key in got, 'the oracle produces no %s' % key
        _close(got[key], exp, fx.tol(key), '%s %s' % (fx.name, key))
        checked += 1
    assert checked >= 6
    # regulariser-only variables: gradient == weight_decay * value (the generator asserted it on the reference side)
    for vn in fx.meta['reg_only_grad']:
        _close(got['grad/var/' + vn], fx.meta['weight_decay'] * fx.variables[vn], 1e-12, vn)


def test_reference_graph_facts_the_fixtures_expose():
    """Structure the reference's own code decides, read off the fixtures' metadata."""
    m2 = rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_cfg002_train.npz')).meta
    # cfg 002: the PoseLogits convs exist and are regularised although nothing consumes them
    assert m2['var_order'][:4] == ['PoseLogits/ExtraConv2d_1x1/weights', 'PoseLogits/ExtraConv2d_1x1/biases',
                                   'PoseLogits/Conv2d_1c_1x1/weights', 'PoseLogits/Conv2d_1c_1x1/biases']
    assert m2['reg_only_grad'] == ['PoseLogits/ExtraConv2d_1x1/weights', 'PoseLogits/Conv2d_1c_1x1/weights']
    assert m2['n_reg_losses'] == 4 and m2['n_losses'] == 1 and m2['weight_decay'] == 0.0005
    assert m2['draws'] == [{'kind': 'dropout', 'keep_prob': 0.2, 'shape': [3, 5, 5, 128]}]   # DROPOUT -1 -> keep 0.2
    assert m2['backbone_call'] == {'num_classes': 393, 'is_training': True, 'train_top_bn': False, 'kwargs': {}}
    m3 = rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_cfg003_train.npz')).meta
    assert m3['n_losses'] == 2 and m3['reg_only_grad'] == []                                  # pose loss, then action loss
    assert m3['net']['USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT'] is False
    mh = rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_cfg002_dropout_half.npz')).meta
    assert mh['backbone_call']['kwargs'] == {'dropout_keep_prob': 0.5} and mh['draws'][0]['keep_prob'] == 0.5
    # rank 3: chained attention convs Conv2d_PrePose_Attn, ..1, ..2 and top-down convs Conv, Conv_1, Conv_2
    mr = rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_rank3_relu_train.npz'))
    names = [v for v in mr.meta['var_order'] if v.startswith(rf.PRE)]
    assert [n.split('/')[1] for n in names[::2]] == ['Conv2d_PrePose_Attn', 'Conv2d_PrePose_Attn1',
                                                     'Conv2d_PrePose_Attn2', 'Conv', 'Conv_1', 'Conv_2']
    assert mr.variables[rf.PRE + 'Conv2d_PrePose_Attn1/weights'].shape == (1, 1, 1, 1)        # consumes conv 0's map
    assert mr.expected('out/ep/PosePrelogitsBasedAttention').shape == (2, 3, 4, 1, 3)
    # the _2LAYER conv under the four arg-scopes
    two = lambda n: [v[len(rf.PRE):] for v in rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_%s.npz' % n)).meta['var_order']
                     if v.startswith(rf.PRE + 'Conv/')]
    assert two('posefeat_2layer_train') == ['Conv/weights', 'Conv/BatchNorm/beta', 'Conv/BatchNorm/gamma',
                                            'Conv/BatchNorm/moving_mean', 'Conv/BatchNorm/moving_variance']
    assert two('inceptionv3_posefeat_2layer_train') == ['Conv/weights', 'Conv/BatchNorm/beta',
                                                        'Conv/BatchNorm/moving_mean', 'Conv/BatchNorm/moving_variance']
    assert two('vgg16_posefeat_2layer_train') == ['Conv/weights', 'Conv/biases']
    assert two('tsn_posefeat_2layer_train') == ['Conv/weights', 'Conv/biases']
    # temporal attention: its conv is regularised too, bias initialised to 1/F
    mt = rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_video_temporal_att_train.npz')).meta
    assert mt['var_init']['TemporalAttention/Conv/biases'] == {'kind': 'constant', 'value': 1.0 / 3}
    assert mt['n_reg_losses'] == 5


build_head, module_tf_names = rf.build_head, rf.module_tf_names


@pytest.mark.parametrize('path', HEAD_PATHS, ids=rf.case_id)
def test_product_variable_table_matches_the_reference_graph(path):
    """names, shapes, the trainable set and the regularised set of the product's head == the variables the
    reference's graph construction created for the same flags."""
    from attentionalpoolingaction_amd import config as apa_config
    fx = rf.HeadFixture(path)
    network_fn, _ = build_head(fx)
    try:
        table = module_tf_names(network_fn)
        assert set(table) == set(fx.meta['var_order']), (sorted(set(table) ^ set(fx.meta['var_order'])))
        for vn in fx.meta['var_order']:
            ref_shape = tuple(fx.variables[vn].shape)
            ref_shape = ref_shape[-2:] if len(ref_shape) == 4 else ref_shape          # [1,1,Cin,Cout] -> [Cin,Cout]
            assert tuple(table[vn].shape) == ref_shape, vn
            assert bool(table[vn].requires_grad) == (vn in fx.meta['trainable']), vn
        reg = {id(w) for w in network_fn.regularized_weights()}
        want = {vn for vn in fx.meta['var_order'] if vn.endswith('/weights')}        # slim.l2_regularizer: conv weights
        assert {vn for vn, t in table.items() if id(t) in reg} == want
    finally:
        apa_config.reset_cfg()


def test_product_initialisers_match_the_reference_table():
    """stddev / constants of the product's parameter initialisation vs the initialisers the reference passed to
    slim.conv2d (recorded by the generator): N(0, 1e-3) everywhere except PoseLogits/Conv2d_1c_1x1 (the resnet
    arg-scope's variance_scaling_initializer) and zero biases."""
    from attentionalpoolingaction_amd import config as apa_config
    fx = rf.HeadFixture(os.path.join(rf.GOLD, 'ref_head_posefeat_2layer_train.npz'))
    init = fx.meta['var_init']
    torch.manual_seed(0)
    network_fn, _ = build_head(fx, in_channels=2048)
    try:
        table = module_tf_names(network_fn)
        for vn, desc in init.items():
            t = table[vn].detach().float()
            if desc['kind'] == 'random_normal':
                assert desc['stddev'] == 0.001 and desc['mean'] == 0.0
                if t.numel() >= 256:
                    assert abs(float(t.std()) / 0.001 - 1.0) < 0.15, vn
                assert float(t.abs().max()) < 0.01, vn
            elif desc['kind'] == 'zeros':
                assert float(t.abs().max()) == 0.0, vn
            elif desc['kind'] == 'ones':
                assert float((t - 1).abs().max()) == 0.0, vn
            elif desc['kind'] == 'variance_scaling':
                assert vn == 'PoseLogits/Conv2d_1c_1x1/weights' and desc['factor'] == 2.0 and desc['mode'] == 'FAN_IN'
                std = (1.3 * 2.0 / 768) ** 0.5                        # truncated normal, the fixture's formula
                assert float(t.abs().max()) <= 2.0 * std * 1.0001 and abs(float(t.std()) / (0.88 * std) - 1.0) < 0.1
            else:
                raise AssertionError('unexpected initialiser %r for %s' % (desc, vn))
    finally:
        apa_config.reset_cfg()


LOSS_CASES = rf.load_loss_cases()


@pytest.mark.parametrize('case', LOSS_CASES, ids=lambda c: c['name'])
def test_oracle_gen_losses_match_reference(case):
    """every branch of src/loss.py:gen_losses executed by the reference code == oracle.gen_losses, values and
    gradients, incl. label resize (TF 1.1 float32 coordinate arithmetic), LOSS_FN_POSE_SAMPLED with the recorded
    tf.random_uniform draws, zero loss weight, all-invalid keypoints."""
    m = case['meta']
    lg = torch.from_numpy(case['logits'].astype(np.float64)).requires_grad_(True)
    Pl = torch.from_numpy(case['Pl'].astype(np.float64)).requires_grad_(True)

    class Cfg(object):
        class TRAIN(object):
            LOSS_FN_POSE_SAMPLED = bool(m.get('sampled', False))
    ep = {}
    if m.get('sampled'):
        ep['PoseLossUniform'] = torch.from_numpy(
            np.stack([case['uniform/%d' % i] for i in range(m['n_draws'])], -1).astype(np.float64))
    losses = orc.gen_losses(torch.from_numpy(case['labels']), lg, m['action'], m['K'], m['awt'],
                            torch.from_numpy(case['lbl'].astype(np.float64)), Pl, m['pose'],
                            torch.from_numpy(case['valid']), m['pwt'], ep, Cfg)
    assert len(losses) == m['n_losses']
    if losses and sum(losses).requires_grad:
        sum(losses).backward()
    _close([float(l.detach()) for l in losses], case['losses'], 1e-12, 'losses')
    _close(lg.grad if lg.grad is not None else torch.zeros_like(lg), case['G'], 1e-12, 'dlogits')
    _close(Pl.grad if Pl.grad is not None else torch.zeros_like(Pl), case['dPl'], 1e-12, 'dPoseLogits')
    if 'PoseLossMask' in case:
        assert np.array_equal(ep['PoseLossMask'].numpy(), case['PoseLossMask'])


@pytest.mark.regen
def test_generator_reproduces_the_committed_fixtures():
    """tests/golden/make_head_reference.py is deterministic: re-running it on the reference tree gives the
    committed arrays AND meta tables back bit for bit (two small head cases + three benchmark-shape cases)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_head_reference',
                                                  os.path.join(rf.GOLD, 'make_head_reference.py'))
    gen = importlib.util.module_from_spec(spec)
    import sys
    saved = dict(sys.modules)
    try:
        spec.loader.exec_module(gen)
        cfgmod, nf, lossmod = gen.load_reference()
        import copy
        defaults = copy.deepcopy(cfgmod.cfg)
        # two small head cases + two benchmark-shape cases (32 x 14 x 14 x 2048 per-class, and round 6's
        # 32 x 15 x 15 x 2048 at the reference's native map; stored by seed / digest)
        for name in ('cfg003_train', 'posefeat_softmax_train', 'perclass_k51_train_baseline_libmask',
                     'cfg002_train_15x15_libmask', 'cfg003_train_15x15_libmask'):    # (the last one: gate-safe inputs)
            case = [c for c in gen.HEAD_CASES + gen.BIG_CASES if c['name'] == name][0]
            out = gen.run_head_case(cfgmod, nf, lossmod, defaults, case)
            d = np.load(os.path.join(rf.GOLD, ('refbig_%s.npz' if case.get('big') else 'ref_head_%s.npz') % name))
            assert set(out) == set(d.files)
            for k in d.files:
                if k == 'meta':
                    assert json.loads(str(out[k])) == json.loads(str(d[k]))
                else:
                    assert np.array_equal(out[k], d[k]), k
    finally:
        for k in list(sys.modules):                 # drop the tensorflow / nets / easydict stand-ins again
            if k not in saved:
                del sys.modules[k]


BIG_PATHS = rf.big_fixture_paths()


@pytest.mark.parametrize('path', BIG_PATHS, ids=rf.case_id)
def test_oracle_matches_reference_at_the_benchmark_shape(path):
    """BASELINE configs[1]-[3] at their real size -- per-GPU batch 32 x 14 x 14 x 2048, K = 393 (cfg 002 training
    and evaluation, cfg 003 training with the pose head and both losses) -- executed by the reference's own
    nets_factory.py / loss.py (make_head_reference.BIG_CASES).  Inputs and large weights come back from their
    seeds, the dropout mask from the library stream's numpy twin; tensors above the storage limit are held as
    whole-tensor projections + 4096 exact samples (tests/golden/apa_digest.py)."""
    fx = rf.HeadFixture(path)
    assert fx.meta['big'] and fx.arrays['in/images'].shape in ((32, 14, 14, 2048), (32, 15, 15, 2048))
    assert fx.meta['num_classes'] == (51 if fx.flag('_PER_CLASS') else 393)          # HMDB-51 per class / MPII
    got = rf.run_oracle(fx)
    keys = fx.output_keys()
    assert 'grad/images' in keys and 'out/logits' in keys and 'out/ep/TopDownAttention' in keys
    for key in keys:
        assert key in got, 'the oracle produces no %s' % key
        # stored-as-float32 tensors: storage rounding; digests / float64 tensors: float64 round-off of sums over
        # 12.8 M terms in a different association order
        fx.check(key, got[key], 1.5e-7 if key in fx.f32_keys else 1e-10, '%s %s' % (fx.name, key), floor=1e-3,
                 tol_proj=1e-9)
    for vn in fx.meta['reg_only_grad']:
        _close(got['grad/var/' + vn], fx.meta['weight_decay'] * fx.variables[vn], 1e-12, vn)
