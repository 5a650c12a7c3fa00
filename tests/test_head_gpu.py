"""GPU parity, second file: committed golden fixtures through the C ABI, the loss kernels, the
torch.autograd head module behind the reference call surface (network_fn / gen_losses), the
reference's own small custom op (zero_out_channels) and the eval consumers."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import attn_pool_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'attn_*.npz'))),
                         ids=lambda p: 'regression_pin_' + os.path.basename(p)[5:-4])
def test_hip_matches_golden_fixture(gpu, path):
    """REGRESSION PINS: attn_*.npz are outputs of this repo's own oracle (tests/golden/make_golden.py), kept so a
    change of results is noticed.  The fixtures produced by the reference's code are ref_head_*.npz
    (tests/test_reference_fixtures_gpu.py)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    d = np.load(path)
    train = bool(d['train'])
    # a training-mode fixture carries its own dropout mask: replayed through APA_FLAG_RNG_EXTERNAL
    seed = cof.pack_keep_mask(torch.from_numpy(d['mask']), device=gpu) if train else 0
    kp = float(d['keep']) if train else 1.0
    fused = 'Xatt' not in d.files
    X = torch.from_numpy(d['X']).to(gpu)
    Xatt = X if fused else torch.from_numpy(d['Xatt']).to(gpu)
    Wa, ba, Wt, bt = (torch.from_numpy(d[k]).to(gpu) for k in ('Wa', 'ba', 'Wt', 'bt'))
    labels = torch.from_numpy(d['labels']).to(gpu)
    flags = cof.attn_flags(bool(d['softmax']), bool(d['relu']), train)
    logits, att, zs, ab, _, ws = cof.attn_pool_fwd(X, Xatt, Wa, ba, Wt, bt, flags=flags, keep_prob=kp, seed=seed)
    loss, G, _, pred = cof.softmax_xent_fwd_bwd(logits, labels, want_pred=True)
    dX, dXatt, dWa, dba, dWt, dbt = cof.attn_pool_bwd(X, Xatt, Wa, ba, Wt, bt, att, zs, ab, G,
                                                      flags=flags, keep_prob=kp, seed=seed, workspace=ws)
    assert np.abs(logits.cpu().numpy() - d['logits']).max() <= 1e-3          # north_star tolerance
    assert _rel(logits.cpu().numpy(), d['logits']) < 2e-5
    assert _rel(att.cpu().numpy().reshape(d['att'].shape), d['att']) < 2e-5
    assert abs(float(loss[0]) - float(d['loss'])) < 2e-5 * abs(float(d['loss']))
    assert np.array_equal(pred.cpu().numpy(), d['logits'].argmax(1))         # bit-exact argmax
    assert _rel(dX.cpu().numpy(), d['dX']) < 5e-5
    assert _rel(dWt.cpu().numpy(), d['dWt']) < 5e-5
    assert _rel(dWa.cpu().numpy(), d['dWa']) < 5e-5
    assert _rel(dbt.cpu().numpy(), d['dbt']) < 5e-5
    if not fused:
        assert _rel(dXatt.cpu().numpy(), d['dXatt']) < 5e-5


def test_loss_kernels_match_golden(gpu):
    """REGRESSION PIN (losses.npz comes from this repo's oracle); the reference-produced loss cases are
    ref_losses.npz (test_hip_gen_losses_match_reference)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    d = np.load(os.path.join(GOLD, 'losses.npz'))
    Pl = torch.from_numpy(d['pose_Pl']).float().to(gpu)
    lbl = torch.from_numpy(d['pose_lbl']).float().to(gpu)
    valid = torch.from_numpy(d['pose_valid']).to(gpu)
    loss, dPl = cof.pose_l2_loss_fwd_bwd(Pl, lbl, valid, wt=float(d['pose_wt']))
    assert abs(float(loss[0]) - float(d['pose_loss'])) < 1e-5 * float(d['pose_loss'])
    assert _rel(dPl.cpu().numpy(), d['pose_dPl']) < 1e-5
    # invalid keypoints carry exactly zero gradient
    inv = ~torch.from_numpy(d['pose_valid'])
    assert float(dPl.cpu()[inv[:, None, None, :].expand_as(dPl.cpu())].abs().max()) == 0.0
    lg = torch.from_numpy(d['xent_logits']).float().to(gpu)
    lab = torch.from_numpy(d['xent_labels']).to(gpu)
    lb, G, probs, pred = cof.softmax_xent_fwd_bwd(lg, lab, wt=float(d['xent_wt']), want_probs=True, want_pred=True)
    assert abs(float(lb[0]) - float(d['xent_loss'])) < 1e-5 * float(d['xent_loss'])
    assert _rel(G.cpu().numpy(), d['xent_G']) < 1e-5
    np.testing.assert_allclose(probs.sum(1).cpu().numpy(), 1.0, atol=1e-6)
    assert np.array_equal(pred.cpu().numpy(), d['xent_logits'].argmax(1))


def test_softmax_xent_argmax_tie_rule_and_grad_scale(gpu):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    lg = torch.zeros(3, 200, device=gpu)
    lg[0, 7] = lg[0, 150] = 2.0          # tie -> first index (np / tf argmax)
    lg[1, 199] = 1.0
    lg[2, :] = -3.0                      # all equal -> 0
    lab = torch.tensor([7, 0, 5], device=gpu)
    _, G1, _, pred = cof.softmax_xent_fwd_bwd(lg, lab, want_pred=True)
    assert pred.tolist() == [7, 199, 0]
    _, G2, _, _ = cof.softmax_xent_fwd_bwd(lg, lab, grad_scale=0.125)     # 1/num_clones of 8 towers
    assert torch.allclose(G2, G1 * 0.125, rtol=0, atol=1e-9)


def test_head_module_autograd_matches_oracle_cfg002(gpu):
    """network_fn / gen_losses call surface (nets_factory.py:94-133, loss.py:4-8) end to end:
    torch.autograd through the HIP Functions == autograd of the CPU restatement."""
    from attentionalpoolingaction_amd import config as apa_config, loss as apa_loss, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'NET': {
        'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
        'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True}})
    network_fn = nets_factory.get_network_fn('resnet_v1_101', 393, 16, cfg, weight_decay=5e-4,
                                             is_training=False, device=gpu)
    head = network_fn.head
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                  # trained-scale weights
        head.att_weights.copy_(torch.randn(2048, 1, generator=g) / 45)
        head.att_biases.copy_(torch.randn(1, generator=g) * 0.1)
        head.td_weights.copy_(torch.randn(2048, 393, generator=g) / 45)
        head.td_biases.copy_(torch.randn(393, generator=g) * 0.1)
    X = torch.relu(torch.randn(3, 15, 15, 2048, generator=g))
    labels = torch.randint(0, 393, (3,), generator=g)
    Xd = X.to(gpu).requires_grad_(True)
    logits, end_points = network_fn(Xd)
    assert set(end_points) >= {'PosePrelogitsBasedAttention', 'Logits'}
    assert end_points['PosePrelogitsBasedAttention'].shape == (3, 15, 15, 1) and logits.shape == (3, 393)
    losses = apa_loss.gen_losses(labels.to(gpu), logits, 'softmax-xentropy', 393, 1.0,
                                 None, None, '', None, 1.0, end_points, cfg)
    total = sum(losses) + apa_loss.l2_regularization(head.regularized_weights(), network_fn.weight_decay)
    total.backward()

    Xr = X.double().requires_grad_(True)
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in head.named_parameters()}
    lr, _ = orc.attentional_pooling(Xr, None, None, [p['att_weights']], [p['att_biases']],
                                    [p['td_weights']], [p['td_biases']], orc.AttnFlags())
    # the regulariser covers the PoseLogits conv weights too, although the pose head is pruned from the data path
    # (REGULARIZATION_LOSSES of the reference graph: tests/golden/ref_head_cfg002_train.npz)
    tr = orc.action_softmax_xent(lr, labels, 393) + orc.l2_regularizer(
        [p['pose_w1'], p['pose_w2'], p['att_weights'], p['td_weights']], 5e-4)
    tr.backward()
    assert abs(float(total) - float(tr)) < 1e-5 * float(tr)
    assert _rel(Xd.grad.cpu().numpy(), Xr.grad.numpy()) < 5e-5
    for k in ('att_weights', 'att_biases', 'td_weights', 'td_biases'):
        assert _rel(getattr(head, k).grad.cpu().numpy(), p[k].grad.numpy()) < 5e-5, k
    # cfg 002: the pose head is pruned from the data path; only the weight decay reaches its weights
    assert torch.allclose(head.pose_w1.grad, 5e-4 * head.pose_w1.detach(), rtol=1e-6, atol=0)
    assert head.pose_b1.grad is None
    apa_config.reset_cfg()


@pytest.mark.parametrize('fuse,softmax', [(True, False), (False, False), (True, True)])
def test_head_module_cfg003_pose_attention_with_pose_loss(gpu, fuse, softmax):
    """cfg 003 through the module surface: PoseLogits end point, attention from pose_pre_logits,
    gen_losses with the pose L2 term; all eight parameter gradients vs the oracle.  fuse=True: pose head
    and pooling as ONE autograd node -- the attention-branch gradient reaches the pose head in rank-1
    form (APA_FLAG_DXATT_RANK1 + apa_pose_head_bwd_rank1ext) and dX is accumulated in place; fuse=False:
    two nodes, the [N,P,768] gradient tensor in between, autograd sums the two dX."""
    from attentionalpoolingaction_amd import config as apa_config, loss as apa_loss, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT': softmax},
                              'TRAIN': {'LOSS_FN_POSE': 'l2'}})
    fn = nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=False, device=gpu,
                                     with_pose_logits=True, fuse_pose_attention=fuse)
    head = fn.head
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for name, prm in head.named_parameters():
            scale = 1.0 / prm.shape[0] ** 0.5 if prm.dim() == 2 else 0.1
            prm.copy_((torch.randn(prm.shape, generator=g) * scale).to(gpu))
    X = torch.relu(torch.randn(2, 7, 7, 2048, generator=g))
    labels = torch.randint(0, 51, (2,), generator=g)
    pose_lbl = torch.rand(2, 7, 7, 16, generator=g)
    valid = torch.rand(2, 16, generator=g) > 0.3
    Xd = X.to(gpu).requires_grad_(True)
    logits, ep = fn(Xd)
    assert ep['PoseLogits'].shape == (2, 7, 7, 16)
    losses = apa_loss.gen_losses(labels.to(gpu), logits, 'softmax-xentropy', 51, 1.0,
                                 pose_lbl.to(gpu), ep['PoseLogits'], 'l2', valid.to(gpu), 1.0, ep, cfg)
    sum(losses).backward()

    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in head.named_parameters()}
    Xr = X.double().requires_grad_(True)
    pre, pl = orc.pose_logits_head(Xr, p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'])
    lr, _ = orc.attentional_pooling(Xr, pre, pl, [p['att_weights']], [p['att_biases']],
                                    [p['td_weights']], [p['td_biases']],
                                    orc.AttnFlags(single_layer_att=False, softmax_att=softmax))
    sum(orc.gen_losses(labels, lr, 'softmax-xentropy', 51, 1.0, pose_lbl.double(), pl, 'l2', valid, 1.0)).backward()
    assert _rel(logits.detach().cpu().numpy(), lr.detach().numpy()) < 2e-5
    assert _rel(Xd.grad.cpu().numpy(), Xr.grad.numpy()) < 1e-4
    for k, v in head.named_parameters():
        if softmax and k == 'att_biases':      # softmax is shift-invariant: this gradient is exactly 0
            assert float(v.grad.abs().max()) < 1e-6
            continue
        assert _rel(v.grad.cpu().numpy(), p[k].grad.numpy()) < 1e-4, k
    apa_config.reset_cfg()


def test_cfg003_fused_node_matches_two_nodes_bf16_and_single_loss_cases(gpu):
    """PoseAttentionFunction against the two separate autograd nodes, bf16 features: same logits
    (identical forward kernels), gradients within bf16 resolution (the rank-1 hand-over skips one bf16
    rounding of the attention-branch gradient); and the two degenerate backward calls -- only the action
    loss (no dPl) and only the pose loss (no dlogits)."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True}, 'TRAIN': {'LOSS_FN_POSE': 'l2'}})
    heads = []
    for fuse in (True, False):
        torch.manual_seed(3)
        heads.append(nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=False, device=gpu,
                                                 with_pose_logits=True, fuse_pose_attention=fuse).head)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for name, prm in heads[0].named_parameters():
            scale = 1.0 / prm.shape[0] ** 0.5 if prm.dim() == 2 else 0.1
            prm.copy_((torch.randn(prm.shape, generator=g) * scale).to(gpu))
    heads[1].load_state_dict(heads[0].state_dict())
    X = torch.relu(torch.randn(3, 7, 7, 2048, generator=g)).bfloat16().to(gpu)
    labels = torch.tensor([1, 7, 30], device=gpu)
    for mode in ('both', 'action', 'pose'):
        grads = []
        for head in heads:
            head.zero_grad()
            Xd = X.clone().requires_grad_(True)
            logits, ep = head(Xd)
            loss = 0.0
            if mode in ('both', 'action'):
                loss = loss + torch.nn.functional.cross_entropy(logits, labels)
            if mode in ('both', 'pose'):
                loss = loss + (ep['PoseLogits'] ** 2).mean()
            loss.backward()
            grads.append((logits.detach(), Xd.grad.float(), {k: (None if v.grad is None else v.grad.clone())
                                                              for k, v in head.named_parameters()}))
        assert torch.equal(grads[0][0], grads[1][0])
        assert _rel(grads[0][1].cpu().numpy(), grads[1][1].cpu().numpy()) < 2e-2, mode
        for k in grads[0][2]:
            a, b = grads[0][2][k], grads[1][2][k]
            assert (a is None) == (b is None), (mode, k)
            if a is not None:
                assert _rel(a.cpu().numpy(), b.cpu().numpy()) < 2e-2, (mode, k)
    apa_config.reset_cfg()


def test_head_module_video_frame_pooling_and_training_mode(gpu):
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT': True}})
    fn = nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=False, device=gpu)
    vid = torch.relu(torch.randn(2, 3, 7, 7, 2048, device=gpu))          # [B, F, H, W, C]
    with torch.no_grad():
        fn.head.td_weights.mul_(20)
        logits, ep = fn(vid)
    assert logits.shape == (2, 51) and ep['logits_beforePool'].shape == (6, 51)
    want, _ = orc.frame_pooling(ep['logits_beforePool'].cpu().double(), 3)
    assert _rel(logits.cpu().numpy(), want.numpy()) < 1e-6
    att = ep['PosePrelogitsBasedAttention']
    np.testing.assert_allclose(att.sum((1, 2, 3)).cpu().numpy(), 1.0, atol=1e-5)   # spatial softmax
    # training mode: dropout active, a new mask every call
    fn_t = nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=True, device=gpu)
    with torch.no_grad():
        fn_t.head.td_weights.mul_(20)
        a, _ = fn_t(vid[:, 0])
        b, _ = fn_t(vid[:, 0])
    assert not torch.equal(a, b)
    apa_config.reset_cfg()


def test_pose_loss_through_gen_losses_with_label_resize(gpu):
    from attentionalpoolingaction_amd import loss as apa_loss
    g = torch.Generator().manual_seed(4)
    Pl = torch.randn(2, 5, 5, 16, generator=g)
    lbl = torch.rand(2, 10, 10, 16, generator=g)          # different size -> TF1 legacy resize
    valid = torch.rand(2, 16, generator=g) > 0.4
    Pd = Pl.to(gpu).requires_grad_(True)
    (lp,) = apa_loss.gen_losses(None, None, '', 0, 1.0, lbl.to(gpu), Pd, 'l2', valid.to(gpu), 2.0)
    lp.backward()
    Pr = Pl.double().requires_grad_(True)
    want = orc.pose_l2_loss(Pr, lbl.double(), valid, 2.0)
    want.backward()
    assert abs(float(lp) - float(want)) < 1e-5 * float(want)
    assert _rel(Pd.grad.cpu().numpy(), Pr.grad.numpy()) < 1e-5


@pytest.mark.parametrize('shape', [((2, 10, 10, 16), (5, 5)), ((3, 15, 15, 16), (14, 14)), ((1, 7, 9, 3), (15, 15)),
                                   ((2, 14, 14, 16), (15, 15))])
def test_label_resize_kernel_matches_tf1_legacy_rule(gpu, shape):
    """apa_resize_bilinear_tf1 (gather form, f32) vs the oracle's float64 restatement of TF 1.1's
    legacy bilinear rule (Appendix B): down- and up-sampling, non-square, the clamped last row / column."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    (n, h, w, c), (oh, ow) = shape
    img = torch.rand(n, h, w, c, generator=torch.Generator().manual_seed(h * 31 + ow))
    got = cof.resize_bilinear_tf1(img.to(gpu), oh, ow).cpu()
    want = orc.tf1_resize_bilinear(img.double(), oh, ow)
    assert got.shape == (n, oh, ow, c)
    assert float((got.double() - want).abs().max()) < 2e-6


@pytest.mark.parametrize('kind', ['l2', 'multi-label', 'multi-label-2'])
def test_other_action_losses_through_gen_losses(gpu, kind):
    """src/loss.py:81-101 through the 12-argument gen_losses: value and gradient vs the oracle."""
    from attentionalpoolingaction_amd import loss as apa_loss
    g = torch.Generator().manual_seed(21)
    N, K, wt = 9, 157, 0.7
    logits = torch.randn(N, K, generator=g) * 2.5
    if kind == 'l2':
        labels = torch.randint(0, K, (N,), generator=g)
    else:
        labels = (torch.rand(N, K, generator=g) < 0.08).long()       # multi-hot, int like the tfrecord field
    Ld = logits.to(gpu).requires_grad_(True)
    (la,) = apa_loss.gen_losses(labels.to(gpu), Ld, kind, K, wt, None, None, '', None, 1.0)
    la.backward()
    Lr = logits.double().requires_grad_(True)
    if kind == 'l2':
        want = orc.action_l2(Lr, labels, K, wt)
    elif kind == 'multi-label':
        want = orc.action_multi_label(Lr, labels)                     # the reference ignores the weight here
    else:
        want = orc.action_multi_label_2(Lr, labels)
    want.backward()
    assert abs(float(la) - float(want)) < 2e-6 * max(1.0, abs(float(want)))
    assert _rel(Ld.grad.cpu().numpy(), Lr.grad.numpy()) < 5e-6


def test_sampled_pose_loss_through_gen_losses(gpu):
    """cfg.TRAIN.LOSS_FN_POSE_SAMPLED (src/loss.py:36-52) with the label resize in front: value, gradient and
    the PoseLossMask end point vs the literal oracle, fed the same uniform draws.  Labels are sparse
    blobs (zeros elsewhere) like real heat-maps, so both the "selected negatives" and the "positive
    logits" parts of the mask are populated."""
    from attentionalpoolingaction_amd import config as apa_config, loss as apa_loss
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'TRAIN': {'LOSS_FN_POSE_SAMPLED': True}})
    g = torch.Generator().manual_seed(8)
    N, H, J = 4, 15, 16
    Pl = torch.randn(N, H, H, J, generator=g) * 0.5
    lbl = torch.rand(N, H, H, J, generator=g)
    lbl = torch.where(lbl > 0.8, lbl, torch.zeros_like(lbl))
    valid = torch.rand(N, J, generator=g) > 0.3
    u = torch.rand(N, H, H, J, generator=g)
    Pd = Pl.to(gpu).requires_grad_(True)
    ep = {'PoseLossUniform': u.to(gpu)}
    (lp,) = apa_loss.gen_losses(None, None, '', 0, 1.0, lbl.to(gpu), Pd, 'l2', valid.to(gpu), 3.0, ep, cfg)
    lp.backward()
    Pr = Pl.double().requires_grad_(True)
    want, wmask = orc.pose_l2_sampled_loss(Pr, lbl.double(), valid, u.double(), 3.0)
    want.backward()
    assert torch.equal(ep['PoseLossMask'].cpu().double(), wmask)
    assert 0.2 < float(wmask.mean()) < 0.9
    assert abs(float(lp) - float(want)) < 2e-6 * float(want)
    assert _rel(Pd.grad.cpu().numpy(), Pr.grad.numpy()) < 5e-6
    apa_config.reset_cfg()


def test_zero_out_channels_reference_case(gpu):
    """src/custom_ops/test/zero_out_channels_op_test.py:10-18: ones((1,3,3,5)), mask [T,F,T,T,T]
    -> channel 1 zero, the others one."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    x = torch.ones(1, 3, 3, 5, device=gpu)
    out = cof.zero_out_channels(x, torch.tensor([True, False, True, True, True], device=gpu))
    assert float(out[..., 1].abs().max()) == 0.0
    assert float((out[..., [0, 2, 3, 4]] - 1).abs().max()) == 0.0


def test_eval_consumer_predict_and_map(gpu):
    from attentionalpoolingaction_amd import eval_utils
    from oracle import labels_eval_oracle as leo
    g = torch.Generator().manual_seed(6)
    logits = torch.randn(64, 393, generator=g) * 3
    labels = torch.randint(0, 393, (64,), generator=g).numpy()
    probs, pred = eval_utils.predict(logits.to(gpu))
    sm, acc, mAP = leo.eval_consumer(logits.numpy(), labels)
    assert np.array_equal(pred.cpu().numpy(), logits.numpy().argmax(1))
    np.testing.assert_allclose(probs.cpu().numpy(), sm, rtol=2e-5, atol=1e-8)
    got_map = eval_utils.compute_map(probs.cpu().numpy(), labels)[0]
    assert abs(got_map - mAP) < 1e-6
    assert eval_utils.accuracy(probs.cpu().numpy(), labels) == pytest.approx(acc)


def test_temporal_attention_frame_pooling_forward_backward(gpu):
    """cfg.NET.USE_TEMPORAL_ATT (nets_factory.py:362-373): logits * conv1x1(logits; K->1, bias 1/F),
    mean over frames -- HIP forward/backward vs autograd of the oracle."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    g = torch.Generator().manual_seed(12)
    B, F, K = 3, 5, 51
    x = torch.randn(B * F, K, generator=g)
    w = torch.randn(K, 1, generator=g) * 0.1
    b = torch.full((1,), 1.0 / F)
    gout = torch.randn(B, K, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    pooled_r, ep = orc.frame_pooling(xr, F, wr, br)
    (pooled_r * gout.double()).sum().backward()
    pooled, tatt = cof.frame_pool_fwd(x.to(gpu), F, w.view(-1).to(gpu), b.to(gpu))
    assert _rel(pooled.cpu().numpy(), pooled_r.detach().numpy()) < 1e-5
    assert _rel(tatt.cpu().numpy(), ep['TemporalAttention'].detach().reshape(-1).numpy()) < 1e-5
    dx, dw, db = cof.frame_pool_bwd(x.to(gpu), F, w.view(-1).to(gpu), tatt, gout.to(gpu))
    assert _rel(dx.cpu().numpy(), xr.grad.numpy()) < 1e-5
    assert _rel(dw.cpu().numpy(), wr.grad.reshape(-1).numpy()) < 1e-5
    assert _rel(db.cpu().numpy(), br.grad.numpy()) < 1e-5
    # plain mean pooling
    pooled2, none = cof.frame_pool_fwd(x.to(gpu), F)
    assert none is None and _rel(pooled2.cpu().numpy(), x.view(B, F, K).mean(1).numpy()) < 1e-6
    dx2, _, _ = cof.frame_pool_bwd(x.to(gpu), F, None, None, gout.to(gpu))
    assert _rel(dx2.cpu().numpy(), (gout / F).repeat_interleave(F, 0).numpy()) < 1e-6


def test_network_fn_with_temporal_attention_trains(gpu):
    from attentionalpoolingaction_amd import config as apa_config, loss as apa_loss, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True,
                                      'USE_TEMPORAL_ATT': True}})
    fn = nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=False, device=gpu)
    vid = torch.relu(torch.randn(2, 4, 7, 7, 2048, device=gpu)).requires_grad_(True)
    with torch.no_grad():
        fn.head.td_weights.mul_(20)
    logits, ep = fn(vid)
    assert logits.shape == (2, 51) and ep['TemporalAttention'].shape == (2, 4, 1, 1)
    assert float(fn.temporal['biases']) == pytest.approx(0.25)           # 1 / frames_per_video
    (loss,) = apa_loss.gen_losses(torch.tensor([3, 7], device=gpu), logits, 'softmax-xentropy', 51, 1.0,
                                  None, None, '', None, 1.0)
    loss.backward()
    assert vid.grad is not None and fn.temporal['weights'].grad is not None
    assert float(fn.temporal['weights'].grad.abs().max()) > 0
    apa_config.reset_cfg()


def test_m1_topdown_endpoint_dump(gpu):
    """end_points['TopDownAttention'] (nets_factory.py:309) for the factorised path: materialised on
    request only (eval.py --ept), must equal X.Wt + bt."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    g = torch.Generator().manual_seed(2)
    X = torch.relu(torch.randn(2, 7, 7, 512, generator=g))
    Wa = torch.randn(512, 1, generator=g) / 22; ba = torch.zeros(1)
    Wt = torch.randn(512, 51, generator=g) / 22; bt = torch.randn(51, generator=g) * 0.1
    d = lambda t: t.to(gpu)
    Xd = d(X)
    logits, att, zs, ab, td, ws = cof.attn_pool_fwd(Xd, Xd, d(Wa), d(ba), d(Wt), d(bt), want_topdown=True)
    want = X.double() @ Wt.double() + bt.double()
    assert td.shape == (2, 49, 51)
    assert _rel(td.cpu().numpy().reshape(2, 7, 7, 51), want.numpy()) < 2e-5
    # and the logits are its attention-weighted spatial mean (the literal reference formula)
    lit = (att.cpu().double().view(2, 7, 7, 1) * want).mean((1, 2))
    assert _rel(logits.cpu().numpy(), lit.numpy()) < 2e-5


def _make_head(gpu, net_flags, is_training=False, **kw):
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    net = {'USE_POSE_PRELOGITS_BASED_ATTENTION': True}
    net.update(net_flags)
    apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'NET': net})
    fn = nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=is_training, device=gpu, **kw)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():                                  # trained-scale weights everywhere
        for name, p in fn.head.named_parameters():
            if p.dim() >= 2 and p.shape[0] > 1:
                p.copy_(torch.randn(p.shape, generator=g) / p.shape[0] ** 0.5)
            elif p.dim() >= 2:                             # the [1,1] chained attention convs
                p.copy_(torch.randn(p.shape, generator=g) * 0.5 + 1.0)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return fn, cfg


@pytest.mark.parametrize('single_layer', [True, False])
def test_head_module_rank3_matches_oracle(gpu, single_layer):
    """..._RANK = 3 (nets_factory.py:258-274, 298-309, 322-328): chained [1,1] attention convs, one
    top-down conv per rank, sum over ranks -- two HIP streaming passes via the affine collapse --
    against the literal stacked formulation of the oracle, values and every gradient."""
    from attentionalpoolingaction_amd import config as apa_config
    fn, _ = _make_head(gpu, {'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': single_layer,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_RANK': 3})
    head = fn.head
    g = torch.Generator().manual_seed(9)
    X = torch.relu(torch.randn(3, 14, 14, 2048, generator=g))
    labels = torch.randint(0, 51, (3,), generator=g)
    Xd = X.to(gpu).requires_grad_(True)
    logits, ep = fn(Xd)
    assert ep['PosePrelogitsBasedAttention'].shape == (3, 14, 14, 1, 3)
    loss = torch.nn.functional.cross_entropy(logits, labels.to(gpu))
    loss.backward()

    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in head.named_parameters()}
    Xr = X.double().requires_grad_(True)
    pre = None
    if not single_layer:
        pre, _ = orc.pose_logits_head(Xr, p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'])
    aw = [p['att_weights']] + [p['att_weights_r.%d' % r] for r in range(2)]
    ab = [p['att_biases']] + [p['att_biases_r.%d' % r] for r in range(2)]
    tw = [p['td_weights']] + [p['td_weights_r.%d' % r] for r in range(2)]
    tb = [p['td_biases']] + [p['td_biases_r.%d' % r] for r in range(2)]
    lr, epr = orc.attentional_pooling(Xr, pre, None, aw, ab, tw, tb,
                                      orc.AttnFlags(single_layer_att=single_layer, rank=3))
    torch.nn.functional.cross_entropy(lr, labels).backward()
    assert _rel(logits.detach().cpu().numpy(), lr.detach().numpy()) < 2e-5
    assert _rel(ep['PosePrelogitsBasedAttention'].detach().cpu().numpy(),
                epr['PosePrelogitsBasedAttention'].detach().numpy()) < 2e-5
    assert _rel(Xd.grad.cpu().numpy(), Xr.grad.numpy()) < 1e-4
    for k, v in head.named_parameters():
        if p[k].grad is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        assert _rel(v.grad.cpu().numpy(), p[k].grad.numpy()) < 1e-4, k
    assert len(head.tf_variable_names()) == 8 + 4 * 2
    apa_config.reset_cfg()


@pytest.mark.parametrize('single_layer,relu,per_class,topdown,pose_feat,train', [
    (True, True, False, False, False, False),     # relu on every stacked map
    (False, False, True, False, False, True),     # per-class maps from pose_pre_logits, [K,K] chained convs
    (True, True, True, False, False, False),
    (True, False, False, True, False, False),     # TopDownAttention dump [N,H,W,K,R]
    (False, False, False, False, True, True),     # _WITH_POSE_FEAT: every rank's top-down conv sees C + J
    (True, True, True, True, True, True),         # per-class maps + pose features (+ dump): literal concatenation
    (False, False, False, True, True, False),     # class-agnostic map + pose features + dump: same route
])
def test_head_module_rank_gt1_general_forms(gpu, single_layer, relu, per_class, topdown, pose_feat, train):
    """..._RANK > 1 beyond the collapsed identity form: the reference's loop (nets_factory.py:258-274, 298-309,
    322-328) also admits relu, per-class maps, the end-point dump and the pose features; here one pass of the
    HIP op per rank with the chained convs folded into effective attention weights.  (softmax + rank > 1 does
    not build in the reference.)"""
    from attentionalpoolingaction_amd import config as apa_config
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    R = 2 if pose_feat else 3
    kw = {'want_topdown': True} if topdown else {}
    fn, _ = _make_head(gpu, {'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': single_layer,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_RANK': R,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_RELU_ATT': relu,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS': per_class,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT': pose_feat},
                       is_training=train, **kw)
    head = fn.head
    assert not head.rank_collapsed
    with torch.no_grad():      # chained [M,M] convs near the identity, so that every rank matters
        for r in range(R - 1):
            m = head.att_weights_r[r].shape[0]
            head.att_weights_r[r].copy_(torch.eye(m) * 0.8 + torch.randn(m, m, generator=torch.Generator().manual_seed(r)) * 0.05)
    N, H, C, K, J = 2, 7, 2048, 51, 16
    g = torch.Generator().manual_seed(23)
    X = torch.relu(torch.randn(N, H, H, C, generator=g))
    labels = torch.randint(0, K, (N,), generator=g)
    Xd = X.to(gpu).requires_grad_(True)
    step0 = head._step
    logits, ep = fn(Xd)
    torch.nn.functional.cross_entropy(logits, labels.to(gpu)).backward()
    mask = None
    if train:
        ce = C + (J if pose_feat else 0)
        m = cof.dropout_mask((N * H * H * ce,), head.keep_prob, head.seed, step0).cpu()
        mask = m[:N * H * H * C].view(N, H, H, C)
        if pose_feat and (per_class or topdown):   # the op sees the concatenated tensor: one flat stream
            mask = m.view(N, H, H, C + J)
        elif pose_feat:
            mask = torch.cat([mask, m[N * H * H * C:].view(N, H, H, J)], dim=-1)
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in head.named_parameters()}
    Xr = X.double().requires_grad_(True)
    pre = pl = None
    if not single_layer or pose_feat:
        pre, pl = orc.pose_logits_head(Xr, p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'])
    aw = [p['att_weights']] + [p['att_weights_r.%d' % r] for r in range(R - 1)]
    ab = [p['att_biases']] + [p['att_biases_r.%d' % r] for r in range(R - 1)]
    tw = [p['td_weights']] + [p['td_weights_r.%d' % r] for r in range(R - 1)]
    tb = [p['td_biases']] + [p['td_biases_r.%d' % r] for r in range(R - 1)]
    lr, epr = orc.attentional_pooling(
        Xr, pre, pl, aw, ab, tw, tb,
        orc.AttnFlags(single_layer_att=single_layer, rank=R, relu_att=relu, per_class=per_class,
                      with_pose_feat=pose_feat), is_training=train, keep_prob=head.keep_prob, dropout_mask=mask)
    torch.nn.functional.cross_entropy(lr, labels).backward()
    assert _rel(logits.detach().cpu().numpy(), lr.detach().numpy()) < 5e-5
    att_ref = epr['PosePrelogitsBasedAttention']
    assert ep['PosePrelogitsBasedAttention'].shape == att_ref.shape == (N, H, H, K if per_class else 1, R)
    assert _rel(ep['PosePrelogitsBasedAttention'].detach().cpu().numpy(), att_ref.detach().numpy()) < 5e-5
    if topdown:
        assert ep['TopDownAttention'].shape == (N, H, H, K, R)
        assert _rel(ep['TopDownAttention'].detach().cpu().float().numpy(), epr['TopDownAttention'].detach().numpy()) < 5e-5
    assert _rel(Xd.grad.cpu().numpy(), Xr.grad.numpy()) < 2e-4
    for k, v in head.named_parameters():
        if p[k].grad is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        assert _rel(v.grad.cpu().numpy(), p[k].grad.numpy()) < 2e-4, k
    if pose_feat:
        assert head.td_weights_r[0].shape == (C + J, K)
    apa_config.reset_cfg()
    # softmax over the stacked maps does not build in the reference either
    with pytest.raises(ValueError):
        _make_head(gpu, {'USE_POSE_PRELOGITS_BASED_ATTENTION_RANK': 2,
                         'USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT': True})
    apa_config.reset_cfg()


@pytest.mark.parametrize('single_layer,two_layer,train', [(False, False, False), (False, False, True),
                                                          (True, False, True), (False, True, True),
                                                          (True, True, False)])
def test_head_module_with_pose_feat_matches_oracle(gpu, single_layer, two_layer, train):
    """..._WITH_POSE_FEAT[_2LAYER] (nets_factory.py:289-296): the top-down conv sees concat(last_conv,
    pose_logits) -- here through apa_attn_pool_{fwd,bwd}_cat, the concatenation never formed.  Training
    mode hands the oracle the kernel's own keep-mask of the [N,P,2048+16] tensor (X part = the flat
    stream, extra channels = its continuation at N*P*C); the 2-layer variant adds conv + batch-norm
    (batch statistics) + relu in front of the concatenation."""
    from attentionalpoolingaction_amd import config as apa_config
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    fn, _ = _make_head(gpu, {'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': single_layer,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT': True,
                             'USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT_2LAYER': two_layer},
                       is_training=train)
    head = fn.head
    assert head.td_weights.shape == (2048 + 16, 51)
    with torch.no_grad():
        if two_layer:
            head.pose_feat_weights.copy_(torch.randn(16, 16, generator=torch.Generator().manual_seed(3)) * 0.3)
            head.pose_feat_bn_gamma.copy_(torch.rand(16, generator=torch.Generator().manual_seed(4)) + 0.5)
            head.pose_feat_bn_beta.copy_(torch.randn(16, generator=torch.Generator().manual_seed(5)) * 0.2)
    N, H, C, J = 3, 14, 2048, 16
    g = torch.Generator().manual_seed(19)
    X = torch.relu(torch.randn(N, H, H, C, generator=g))
    labels = torch.randint(0, 51, (N,), generator=g)
    Xd = X.to(gpu).requires_grad_(True)
    step0 = head._step
    logits, ep = fn(Xd)
    torch.nn.functional.cross_entropy(logits, labels.to(gpu)).backward()
    mask = None
    if train:
        m = cof.dropout_mask((N * H * H * (C + J),), head.keep_prob, head.seed, step0).cpu()
        mx = m[:N * H * H * C].view(N, H, H, C)
        me = m[N * H * H * C:].view(N, H, H, J)
        mask = torch.cat([mx, me], dim=-1)
        assert abs(float(me.float().mean()) - head.keep_prob) < 0.03
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in head.named_parameters()}
    Xr = X.double().requires_grad_(True)
    pre, pl = orc.pose_logits_head(Xr, p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'])
    lr, _ = orc.attentional_pooling(
        Xr, pre, pl, [p['att_weights']], [p['att_biases']], [p['td_weights']], [p['td_biases']],
        orc.AttnFlags(single_layer_att=single_layer, with_pose_feat=True, with_pose_feat_2layer=two_layer),
        is_training=train, keep_prob=head.keep_prob, dropout_mask=mask,
        pose_feat_w=p.get('pose_feat_weights'),
        pose_feat_bn=(p['pose_feat_bn_gamma'], p['pose_feat_bn_beta']) if two_layer else None)
    torch.nn.functional.cross_entropy(lr, labels).backward()
    assert _rel(logits.detach().cpu().numpy(), lr.detach().numpy()) < 2e-5
    assert _rel(ep['PoseLogits'].detach().cpu().numpy(), pl.detach().numpy()) < 2e-5
    assert _rel(Xd.grad.cpu().numpy(), Xr.grad.numpy()) < 2e-4
    names = ['pose_w1', 'pose_b1', 'pose_w2', 'pose_b2', 'att_weights', 'att_biases', 'td_weights', 'td_biases']
    if two_layer:
        names += ['pose_feat_weights', 'pose_feat_bn_gamma', 'pose_feat_bn_beta']
        tfn = head.tf_variable_names()
        assert tfn['pose_feat_weights'].endswith('Attention/Conv/weights') and tfn['td_weights'].endswith('Conv_1/weights')
    for k in names:
        if two_layer and k == 'pose_b2':
            # batch-norm removes any per-channel shift of its input: d/d(pose_b2) of the attention branch
            # is exactly 0 (only rounding noise on both sides)
            assert float(getattr(head, k).grad.abs().max()) < 1e-6 * float(head.pose_w2.grad.abs().max())
            continue
        assert _rel(getattr(head, k).grad.cpu().numpy(), p[k].grad.numpy()) < 2e-4, k
    assert float(getattr(head, 'td_weights').grad[C:].abs().max()) > 0      # the extra rows do get a gradient
    apa_config.reset_cfg()


def test_separate_pose_tap_of_the_tsn_inception_config(gpu):
    """cfg.NET.LAST_CONV_MAP_FOR_POSE (nets_factory.py:148-150, config.py:218-222): for inception_v2_tsn the
    pose head reads `inception_5a` while the pooled features are `inception_5b`.  A backbone that returns its
    end points by name gets both taps routed; gradients reach each tap separately."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'MODEL_NAME': 'inception_v2_tsn', 'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True}})
    N, H, C, K, J = 3, 7, 1024, 51, 16
    g = torch.Generator().manual_seed(41)
    A = torch.relu(torch.randn(N, H, H, C, generator=g))      # inception_5a: pose tap
    B = torch.relu(torch.randn(N, H, H, C, generator=g))      # inception_5b: pooled features
    Ad, Bd = A.to(gpu).requires_grad_(True), B.to(gpu).requires_grad_(True)
    fn = nets_factory.get_network_fn(
        'inception_v2_tsn', K, J, cfg, is_training=False, device=gpu, with_pose_logits=True,
        backbone=lambda images: {'InceptionV2_TSN/inception_5a': Ad, 'InceptionV2_TSN/inception_5b': Bd})
    head = fn.head
    with torch.no_grad():
        for name, p in head.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) / max(p.shape[0], 1) ** 0.5 if p.dim() >= 2
                    else torch.randn(p.shape, generator=g) * 0.1)
    labels = torch.randint(0, K, (N,), generator=g)
    wpl = torch.randn(N, H, H, J, generator=g)
    logits, ep = fn(torch.zeros(N, 8, 8, 3, device=gpu))
    (torch.nn.functional.cross_entropy(logits, labels.to(gpu)) + (ep['PoseLogits'] * wpl.to(gpu)).sum()).backward()
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in head.named_parameters()}
    Ar, Br = A.double().requires_grad_(True), B.double().requires_grad_(True)
    pre, pl = orc.pose_logits_head(Ar, p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'])
    lr, _ = orc.attentional_pooling(Br, pre, pl, [p['att_weights']], [p['att_biases']], [p['td_weights']],
                                    [p['td_biases']], orc.AttnFlags(single_layer_att=False))
    (torch.nn.functional.cross_entropy(lr, labels) + (pl * wpl.double()).sum()).backward()
    assert _rel(logits.detach().cpu().numpy(), lr.detach().numpy()) < 5e-5
    assert _rel(ep['PoseLogits'].detach().cpu().numpy(), pl.detach().numpy()) < 5e-5
    assert _rel(Ad.grad.cpu().numpy(), Ar.grad.numpy()) < 2e-4 and _rel(Bd.grad.cpu().numpy(), Br.grad.numpy()) < 2e-4
    for k in ('pose_w1', 'pose_w2', 'att_weights', 'td_weights'):
        assert _rel(getattr(head, k).grad.cpu().numpy(), p[k].grad.numpy()) < 2e-4, k
    apa_config.reset_cfg()


@pytest.mark.parametrize('bdtype', [None, torch.bfloat16])
def test_network_fn_with_resnet_backbone_end_to_end(gpu, bdtype):
    """images -> slim ResNet-v1-101 (torch-ROCm, channels-last) -> HIP head.
    The block4 tap must reach the op without a layout copy, logits must equal the oracle head
    applied to the same tap, and gradients must reach the first backbone conv."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'NET': {
        'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
        'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True}})
    torch.manual_seed(0)
    fn = nets_factory.get_network_fn('resnet_v1_101', 393, 16, cfg, is_training=False, device=gpu,
                                     with_backbone=True, backbone_dtype=bdtype)
    head = fn.head
    with torch.no_grad():
        head.att_weights.normal_(0, 1 / 45)
        head.td_weights.normal_(0, 1 / 45)
    images = (torch.rand(2, 224, 224, 3) * 255 - 128).to(gpu).requires_grad_(True)
    tap = fn.backbone(images) if bdtype is None else None
    if tap is not None:
        assert tap.shape == (2, 7, 7, 2048) and tap.is_contiguous() and tap.dtype == torch.float32
    logits, ep = fn(images)
    assert logits.shape == (2, 393) and torch.isfinite(logits).all()
    if bdtype is None:
        X = tap.detach().cpu().double()
        lr, _ = orc.attentional_pooling(X, None, None, [head.att_weights.detach().cpu().double()],
                                        [head.att_biases.detach().cpu().double()],
                                        [head.td_weights.detach().cpu().double()],
                                        [head.td_biases.detach().cpu().double()], orc.AttnFlags())
        assert _rel(logits.detach().cpu().numpy(), lr.numpy()) < 2e-5
    else:
        assert ep['PosePrelogitsBasedAttention'].shape == (2, 7, 7, 1)
    labels = torch.tensor([3, 77], device=gpu)
    torch.nn.functional.cross_entropy(logits.float(), labels).backward()
    g = fn.backbone.conv1.conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert images.grad is not None and float(images.grad.abs().max()) > 0
    apa_config.reset_cfg()


@pytest.mark.parametrize('C,dtype,softmax,train', [(2048, torch.float32, False, True), (2048, torch.float32, True, False),
                                                   (1024, torch.float32, True, True), (4096, torch.float32, False, False),
                                                   (2048, torch.bfloat16, False, True), (2048, torch.bfloat16, True, False)])
def test_relu_input_flag_equals_explicit_relu(gpu, C, dtype, softmax, train):
    """APA_FLAG_RELU_INPUT (second half): the op is handed block4's residual sum
    BEFORE the last ReLU (resnet_v1.py:108-109) and applies max(X, 0) on the fly.  Against the op on
    an explicitly rectified map: every forward output and every parameter gradient bit-identical,
    dX == dX_ref * [Xpre > 0] bit-identical (the ReLU's backward, fused into the dX store)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, K = 5, 30, 51
    g = torch.Generator().manual_seed(C + 7 * int(softmax))
    Xpre = torch.randn(N, P, C, generator=g).to(dtype).to(gpu)          # about half negative
    X = torch.relu(Xpre)
    Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(gpu)
    ba = torch.full((1,), 0.05, device=gpu)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu)
    bt = (torch.randn(K, generator=g) * 0.1).to(gpu)
    G = (torch.randn(N, K, generator=g) / N).to(gpu)
    outs = []
    for x, rin in ((X, False), (Xpre, True)):
        flags = cof.attn_flags(softmax, False, train, relu_input=rin)
        logits, att, zsave, abar, _, ws = cof.attn_pool_fwd(x, x, Wa, ba, Wt, bt, flags=flags, keep_prob=0.5,
                                                            seed=9, offset=3)
        grads = cof.attn_pool_bwd(x, x, Wa, ba, Wt, bt, att, zsave, abar, G, flags=flags, keep_prob=0.5,
                                  seed=9, offset=3, workspace=ws)
        torch.cuda.synchronize()
        outs.append((logits, att, zsave, abar) + tuple(grads))
    ref, got = outs
    for i, name in enumerate(('logits', 'att', 'zsave', 'abar')):
        assert torch.equal(ref[i], got[i]), name
    dX_ref, dX = ref[4], got[4]
    assert torch.equal(dX, torch.where(Xpre > 0, dX_ref, torch.zeros_like(dX_ref)))
    assert float((dX == 0).float().mean()) > 0.3                          # the mask did something
    for i, name in zip((6, 7, 8, 9), ('dWa', 'dba', 'dWt', 'dbt')):
        assert torch.equal(ref[i], got[i]), name


def test_dxatt_rank1_flag_is_rejected_for_per_class_maps(gpu):
    """APA_FLAG_DXATT_RANK1 only exists for one bottom-up map: with M = K the gradient w.r.t. the attention
    input has rank K and the caller's [N*P] buffer would be overrun -- the call must fail loudly."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, C, K = 2, 9, 2048, 8
    X = torch.randn(N, P, C, device=gpu)
    Xatt = torch.randn(N, P, 768, device=gpu)
    Wa, ba = torch.randn(768, K, device=gpu) * 0.03, torch.zeros(K, device=gpu)
    Wt, bt = torch.randn(C, K, device=gpu) * 0.03, torch.zeros(K, device=gpu)
    logits, att, zsave, abar, _, ws = cof.attn_pool_fwd(X, Xatt, Wa, ba, Wt, bt)
    G = torch.randn(N, K, device=gpu)
    with pytest.raises(cof.ApaError):
        cof.attn_pool_bwd(X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, workspace=ws, dxatt_rank1=True)


def test_relu_input_flag_rejects_paths_without_the_fused_kernels(gpu):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, K = 2, 9, 8
    for C, M, sep in ((512, 1, False), (2048, 8, False), (2048, 1, True)):
        X = torch.randn(N, P, C, device=gpu)
        Xatt = torch.randn(N, P, C, device=gpu) if sep else X
        Wa, ba = torch.randn(C, M, device=gpu), torch.zeros(M, device=gpu)
        Wt, bt = torch.randn(C, K, device=gpu), torch.zeros(K, device=gpu)
        with pytest.raises(cof.ApaError):
            cof.attn_pool_fwd(X, Xatt, Wa, ba, Wt, bt, flags=cof.attn_flags(relu_input=True))


def test_backbone_preactivation_tap_matches_explicit_final_relu(gpu):
    """get_network_fn(with_backbone=True, fuse_final_relu=True): the last bottleneck unit returns its
    residual sum, the head applies the ReLU inside the op.  Same weights, same images: logits, attention
    map and the gradients that reach the backbone equal the unfused network's."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'NET': {
        'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
        'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True}})
    fns = []
    for fuse in (False, True):
        torch.manual_seed(0)
        fns.append(nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=False, device=gpu,
                                               with_backbone=True, fuse_final_relu=fuse))
    fns[1].backbone.load_state_dict(fns[0].backbone.state_dict())
    fns[1].head.load_state_dict(fns[0].head.state_dict())
    with torch.no_grad():
        for fn in fns:
            fn.head.att_weights.copy_(torch.linspace(-1, 1, 2048, device=gpu).view(2048, 1) / 45)
            fn.head.td_weights.copy_(torch.sin(torch.arange(2048 * 51, device=gpu).float()).view(2048, 51) / 45)
    assert fns[1].head.can_fuse_input_relu()
    images = (torch.rand(2, 128, 128, 3, generator=torch.Generator().manual_seed(1)) * 255 - 128).to(gpu)
    res = []
    for fn in fns:
        fn.backbone.zero_grad()
        logits, ep = fn(images)
        torch.nn.functional.cross_entropy(logits, torch.tensor([3, 40], device=gpu)).backward()
        res.append((logits.detach().clone(), ep['PosePrelogitsBasedAttention'].detach().clone(),
                    fn.backbone.blocks[3][2].conv3.conv.weight.grad.clone(), fn.backbone.conv1.conv.weight.grad.clone()))
    # two separately built backbones: MIOpen may pick different conv algorithms, so not bit-level here
    # (the op-level test above is); 1e-5 relative on logits and attention map, 2e-3 on the weight
    # gradients (observed run-to-run: 1e-5 .. 2e-4 on conv1 after 101 layers of MIOpen backward kernels,
    # with identical logits -- the noise is in the backbone's own backward, not in the op)
    for i in range(4):
        assert _rel(res[1][i].cpu().numpy(), res[0][i].cpu().numpy()) < (1e-5 if i < 2 else 2e-3), i
    apa_config.reset_cfg()


def test_pose_logits_fetched_under_the_fused_input_relu_read_the_rectified_map(gpu):
    """with the backbone's last ReLU folded into the op (preactivation=True on a cfg-002-style
    head) `last_conv` is the PRE-activation sum.  end_points['PoseLogits'] is built lazily there; the
    reference's PoseLogits convs read the rectified block4 end point (nets_factory.py:147-160 on
    resnet_v1.py:108's output; src/train.py:400 fetches it unconditionally), so the fetched value and its
    gradient must be those of relu(map) -- through .items() / .values() as well as by key."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'NET': {
        'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
        'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True}})
    torch.manual_seed(3)
    head = nets_factory.AttentionalPoolingHead(51, cfg, in_channels=2048, is_training=False).to(gpu)
    assert head.can_fuse_input_relu()
    with torch.no_grad():
        head.pose_w1.copy_(torch.randn(2048, 768, device=gpu) / 45)
        head.pose_b1.copy_(torch.randn(768, device=gpu) * 0.1)
        head.pose_w2.copy_(torch.randn(768, 16, device=gpu) / 28)
        head.att_weights.copy_(torch.randn(2048, 1, device=gpu) / 45)
        head.td_weights.copy_(torch.randn(2048, 51, device=gpu) / 45)
    pre = torch.randn(2, 5, 5, 2048, device=gpu).requires_grad_(True)      # about half the entries negative
    logits, ep = head(pre, preactivation=True)
    assert 'PoseLogits' in ep and 'PoseLogits' in dict(ep.items()) and len(ep.values()) == len(ep)
    pl = ep['PoseLogits']
    pl.square().sum().backward()
    Xr = torch.relu(pre.detach().double().cpu()).requires_grad_(True)
    _, pl_ref = orc.pose_logits_head(Xr, head.pose_w1.detach().double().cpu(), head.pose_b1.detach().double().cpu(),
                                     head.pose_w2.detach().double().cpu(), head.pose_b2.detach().double().cpu())
    pl_ref.square().sum().backward()
    assert _rel(pl.detach().cpu().numpy(), pl_ref.detach().numpy()) < 2e-5
    g_ref = (Xr.grad * (pre.detach().double().cpu() > 0)).numpy()
    assert _rel(pre.grad.cpu().numpy(), g_ref) < 2e-5
    # and the logits are those of the rectified map too (the op's own fused ReLU)
    lr_, _ = orc.attentional_pooling(torch.relu(pre.detach().double().cpu()), None, None,
                                     [head.att_weights.detach().double().cpu()], [head.att_biases.detach().double().cpu()],
                                     [head.td_weights.detach().double().cpu()], [head.td_biases.detach().double().cpu()],
                                     orc.AttnFlags())
    assert _rel(logits.detach().cpu().numpy(), lr_.numpy()) < 2e-5
    apa_config.reset_cfg()


def test_fused_momentum_sgd_matches_torch_sgd(gpu):
    """apa_momentum_sgd_step == tf.train.MomentumOptimizer + slim L2 on weights only
    (src/train.py:90-94, resnet_utils.py:241) == torch.optim.SGD(momentum, per-group weight_decay);
    three steps, odd sizes (flat offsets are not 16-byte aligned), grad_scale = 1/ITER_SIZE."""
    from attentionalpoolingaction_amd import deploy
    g = torch.Generator().manual_seed(4)
    shapes = {'att_weights': (2048, 1), 'att_biases': (1,), 'td_weights': (2048, 393), 'td_biases': (393,)}
    params = {k: torch.randn(s, generator=g).to(gpu) for k, s in shapes.items()}
    ref = {k: v.detach().cpu().double().requires_grad_(True) for k, v in params.items()}
    opt_ref = torch.optim.SGD([
        {'params': [ref['att_weights'], ref['td_weights']], 'weight_decay': 5e-4},
        {'params': [ref['att_biases'], ref['td_biases']], 'weight_decay': 0.0}], lr=1e-3, momentum=0.9)
    bucket = deploy.GradientBucket(shapes, gpu)
    opt = deploy.MomentumSGD(params, bucket, lr=1e-3, momentum=0.9, weight_decay=5e-4,
                             regularized=['att_weights', 'td_weights'])
    for step in range(3):
        lr = deploy.exponential_decay_lr(1e-3, step, 2, 0.33)
        for k in shapes:
            gk = torch.randn(shapes[k], generator=g)
            bucket.views[k].copy_(gk.to(gpu) * 2.0)            # accumulated over ITER_SIZE = 2
            ref[k].grad = gk.double()
        for grp in opt_ref.param_groups:
            grp['lr'] = lr
        opt_ref.step()
        opt.step(lr=lr, grad_scale=0.5)
    for k in shapes:
        assert _rel(params[k].cpu().numpy(), ref[k].detach().numpy()) < 1e-6, k


@pytest.mark.parametrize('kind', ['adam', 'rmsprop'])
def test_fused_adam_and_rmsprop_match_the_host_rule(gpu, kind):
    """apa_adam_step / apa_rmsprop_step (TRAIN.OPTIMIZER 'adam' / 'rmsprop', src/train.py:84-100): the fused launch
    against deploy's CPU-tensor path of the same TF 1.1 rule (held to a float64 restatement in
    tests/test_deploy_gloo_cpu.py), odd sizes, L2 on the weights only, grad_scale, a bf16 shadow, the reference's
    epsilon = 1.0 and a small one."""
    from attentionalpoolingaction_amd import deploy
    g = torch.Generator().manual_seed(6)
    shapes = {'att_weights': (2048, 1), 'att_biases': (1,), 'td_weights': (2048, 51), 'odd': (1031,)}
    for eps in (1.0, 1e-6):
        cpu = {k: torch.randn(s_, generator=g) for k, s_ in shapes.items()}
        dev = {k: v.clone().to(gpu) for k, v in cpu.items()}
        shadow = {'odd': torch.zeros(1031, dtype=torch.bfloat16, device=gpu)}
        bc, bd = deploy.GradientBucket(shapes, 'cpu'), deploy.GradientBucket(shapes, gpu)
        kw = dict(weight_decay=5e-4, regularized=['att_weights', 'td_weights'])
        if kind == 'adam':
            oc = deploy.Adam(cpu, bc, lr=1e-2, epsilon=eps, **kw)
            od = deploy.Adam(dev, bd, lr=1e-2, epsilon=eps, bf16_shadows=shadow, **kw)
        else:
            oc = deploy.RMSProp(cpu, bc, lr=1e-2, decay=0.9, momentum=0.9, epsilon=eps, **kw)
            od = deploy.RMSProp(dev, bd, lr=1e-2, decay=0.9, momentum=0.9, epsilon=eps, bf16_shadows=shadow, **kw)
        assert torch.equal(shadow['odd'], dev['odd'].to(torch.bfloat16))        # current from the start
        for step in range(4):
            grad = torch.randn(bc.flat.numel(), generator=g)
            bc.flat.copy_(grad)
            bd.flat.copy_(grad.to(gpu))
            oc.step(grad_scale=0.5)
            od.step(grad_scale=0.5)
        for k in shapes:
            assert _rel(dev[k].cpu().numpy(), cpu[k].numpy()) < 2e-6, (kind, eps, k)
        assert torch.equal(shadow['odd'], dev['odd'].to(torch.bfloat16))


def test_momentum_sgd_keeps_a_bf16_shadow_of_the_pose_head_weights_current(gpu):
    """apa_momentum_sgd_step_shadow (deploy.MomentumSGD(bf16_shadows=...)): the optimizer's own launch rewrites the bf16
    operand copy of chosen segments from the UPDATED weights -- bit-identical to rounding the updated fp32 weights, at odd
    sizes (vector body + scalar tail), with the other segments and the update itself untouched; and the one-call cfg 003
    step fed that shadow gives exactly the results of the call that converts W1 itself."""
    from attentionalpoolingaction_amd import deploy
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    g = torch.Generator().manual_seed(9)
    shapes = {'pose_w1': (2048, 768), 'pose_b1': (768,), 'odd': (1031,), 'td_weights': (2048, 51)}
    params = {k: torch.randn(s, generator=g).to(gpu) for k, s in shapes.items()}
    twin = {k: v.clone() for k, v in params.items()}
    bucket, bucket2 = deploy.GradientBucket(shapes, gpu), deploy.GradientBucket(shapes, gpu)
    shadows = {'pose_w1': torch.zeros(2048, 768, dtype=torch.bfloat16, device=gpu),
               'odd': torch.zeros(1031, dtype=torch.bfloat16, device=gpu)}
    opt = deploy.MomentumSGD(params, bucket, lr=1e-2, momentum=0.9, weight_decay=5e-4, regularized=['pose_w1'],
                             bf16_shadows=shadows)
    ref = deploy.MomentumSGD(twin, bucket2, lr=1e-2, momentum=0.9, weight_decay=5e-4, regularized=['pose_w1'])
    for step in range(2):
        grad = torch.randn(bucket.flat.numel(), generator=g).to(gpu)
        bucket.flat.copy_(grad)
        bucket2.flat.copy_(grad)
        opt.step()
        ref.step()
        torch.cuda.synchronize()
        for k in shapes:
            assert torch.equal(params[k], twin[k]), k                          # the update is the same launch body
        for k, sh in shadows.items():
            assert torch.equal(sh, params[k].to(torch.bfloat16)), k            # round-to-nearest-even of the new weights
    # the shadow as the one-call step's W1 operand == the in-call conversion
    N, P, C, Cp, J, K = 2, 25, 2048, 768, 16, 51
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(torch.bfloat16).to(gpu)
    mk = lambda *s_: (torch.randn(*s_, generator=g) / 30).to(gpu)
    W2, b2, Wa, ba, bt = mk(Cp, J), mk(J), mk(Cp, 1), mk(1), mk(K)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    lbl = torch.rand(N, P, J, generator=g).to(gpu)
    valid = (torch.rand(N, J, generator=g) > 0.3).to(gpu)
    outs = []
    for shadow in (None, shadows['pose_w1']):
        e = lambda t: torch.zeros_like(t)
        prm = (params['pose_w1'], params['pose_b1'], W2, b2, Wa, ba, params['td_weights'], bt)
        grads = (e(X),) + tuple(e(t) for t in prm)
        st = cof.PoseAttnTrainStep(X, prm, labels, lbl, valid, grads, flags=cof.attn_flags(False, False, True),
                                   keep_prob=0.5, seed=5, offset=1, w1_bf16=shadow)
        st.run()
        torch.cuda.synchronize()
        outs.append((st.logits.clone(), st.Pl.clone(), grads[0].clone(), grads[1].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_direct_rccl_allreduce_single_rank(gpu):
    """rccl.RcclCommunicator: ctypes ncclCommInitRank / ncclAllReduce on the compute stream.  One GPU
    per box here, so a 1-rank communicator: the sum over ranks is the identity; the N > 1 arithmetic
    (loss / num_clones, regulariser once) is covered by tests/test_deploy_gloo_cpu.py."""
    from attentionalpoolingaction_amd import deploy, rccl
    comm = rccl.RcclCommunicator(0, 1, gpu)
    bucket = deploy.GradientBucket({'w': (2048, 393), 'b': (393,)}, gpu)
    bucket.flat.copy_(torch.arange(bucket.flat.numel(), device=gpu, dtype=torch.float32) % 97)
    want = bucket.flat.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_(bucket.flat, side)
    side.synchronize()
    assert torch.equal(bucket.flat, want)
    cfg2 = deploy.DeploymentConfig(num_clones=2, clone_index=0)
    assert deploy.sum_clone_gradients(bucket, cfg2, comm=comm) is None      # in-stream path
    torch.cuda.synchronize()
    assert torch.equal(bucket.flat, want)
    comm.close()


@pytest.mark.parametrize('N,K', [(6, 51), (32, 393), (33, 512), (64, 129), (1, 4), (5, 600), (70, 51), (300, 393)])
def test_head_train_step_single_call_equals_the_three_entry_points(gpu, N, K):
    """apa_attn_head_train_step (one foreign call per step, cof.HeadTrainStep) against
    apa_attn_pool_fwd + apa_softmax_xent_fwd_bwd + apa_attn_pool_bwd: bit-identical outputs, for both
    feature dtypes, with the device-side dropout counter advancing per step.  K <= 512
    takes the folded loss (logits reduction + row cross-entropy in one kernel, batch mean in the
    backward head kernel: same reduction trees, also for N > 64); (5, 600) the plain three-call sequence."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    P, C = 49, 2048
    for dtype in (torch.float32, torch.bfloat16):
        g = torch.Generator().manual_seed(3)
        X = torch.relu(torch.randn(N, P, C, generator=g)).to(dtype).to(gpu)
        Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(gpu)
        ba = torch.full((1,), 0.1, device=gpu)
        Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu)
        bt = (torch.randn(K, generator=g) * 0.1).to(gpu)
        labels = torch.randint(0, K, (N,), generator=g).to(gpu)
        flags = cof.attn_flags(True, False, True)

        def grads():
            return (torch.empty_like(X), None, torch.empty_like(Wa), torch.empty_like(ba),
                    torch.empty_like(Wt), torch.empty_like(bt))
        ctr_a = torch.zeros(1, dtype=torch.int64, device=gpu)
        ga = grads()
        st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, ga, flags=flags, keep_prob=0.5, seed=7,
                               offset=ctr_a, grad_scale=0.5)
        ctr_b = torch.zeros(1, dtype=torch.int64, device=gpu)
        gb = grads()
        for step in range(2):
            st.run()
            logits, att, zsave, abar, _, ws = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, flags=flags, keep_prob=0.5,
                                                                seed=7, offset=ctr_b)
            loss, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels, grad_scale=0.5)
            cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, att, zsave, abar, G, flags=flags, keep_prob=0.5, seed=7,
                              offset=ctr_b, workspace=ws, out=gb)
            torch.cuda.synchronize()
            assert int(ctr_a.item()) == int(ctr_b.item()) == step + 1
            for a, b, name in ((st.logits, logits, 'logits'), (st.att, att, 'att'), (st.loss, loss, 'loss'),
                               (st.G, G, 'G'), (ga[0], gb[0], 'dX'), (ga[2], gb[2], 'dWa'), (ga[3], gb[3], 'dba'),
                               (ga[4], gb[4], 'dWt'), (ga[5], gb[5], 'dbt')):
                assert torch.equal(a, b), (dtype, step, name)
    with pytest.raises(cof.ApaError):
        cof.HeadTrainStep(X.cpu(), X.cpu(), Wa, ba, Wt, bt, labels, ga)


def test_head_eval_step_single_call(gpu):
    """apa_attn_head_eval_step (cof.HeadEvalStep) == apa_attn_pool_fwd (is_training=False) followed by
    the softmax / argmax consumer of eval.py:193-197, bit for bit; with and without ground truth; a
    TRAIN flag passed by mistake is ignored (evaluation never drops features)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, C, K = 7, 49, 2048, 393
    g = torch.Generator().manual_seed(5)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(gpu)
    Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(gpu)
    ba = torch.zeros(1, device=gpu)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu)
    bt = (torch.randn(K, generator=g) * 0.1).to(gpu)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    flags = cof.attn_flags(True, False, False)
    logits, att, *_ = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, flags=flags)
    loss, _, probs, pred = cof.softmax_xent_fwd_bwd(logits, labels, want_grad=False, want_probs=True, want_pred=True)
    for lab in (labels, None):
        ev = cof.HeadEvalStep(X, X, Wa, ba, Wt, bt, lab, flags=flags | cof.APA_FLAG_TRAIN)
        ev.run()
        torch.cuda.synchronize()
        assert torch.equal(ev.logits, logits) and torch.equal(ev.att, att)
        assert torch.equal(ev.probs, probs) and torch.equal(ev.pred, pred)
        assert torch.equal(ev.pred, logits.argmax(dim=1))
        if lab is not None:
            assert torch.equal(ev.loss, loss)
        else:
            assert ev.loss is None
    # per-class maps go through the same entry
    Wk = (torch.randn(C, 8, generator=g) / C ** 0.5).to(gpu)
    ev = cof.HeadEvalStep(X.bfloat16(), X.bfloat16(), Wk, torch.zeros(8, device=gpu), Wk.clone(),
                          torch.zeros(8, device=gpu))
    xb = X.bfloat16()
    ev = cof.HeadEvalStep(xb, xb, Wk, torch.zeros(8, device=gpu), Wk.clone(), torch.zeros(8, device=gpu))
    ev.run()
    torch.cuda.synchronize()
    assert ev.probs.shape == (N, 8) and torch.allclose(ev.probs.sum(dim=1), torch.ones(N, device=gpu), atol=1e-5)


def test_overlapped_gradient_sum_schedule_matches_the_sequential_loop(gpu):
    """deploy.OverlappedGradientSum: td part reduced + updated on the communication stream between the
    library's grad-ready and td-weights-ready hooks, att part in-stream.  Four SGD steps must be
    bit-identical to the plain sequential loop -- with the communication stream artificially slowed
    down (spin kernel before the td update), so a missing wait in either direction (forward reading
    td_weights early, backward overwriting dWt under the collective) would change the result."""
    from attentionalpoolingaction_amd import deploy, rccl
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, C, K, lr = 8, 49, 2048, 51, 0.05
    g = torch.Generator().manual_seed(11)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(gpu)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    init = [(torch.randn(C, 1, generator=g) / C ** 0.5), torch.zeros(1),
            (torch.randn(C, K, generator=g) / C ** 0.5), torch.zeros(K)]
    flags = cof.attn_flags(False, False, True)
    ws = torch.empty((cof.attn_pool_workspace_bytes(N, P, C, C, K, 1, flags),), dtype=torch.uint8, device=gpu)

    def run(overlapped):
        Wa, ba, Wt, bt = [t.clone().to(gpu) for t in init]
        bucket = torch.zeros(C + 1 + C * K + K, device=gpu)
        b_att, b_td = bucket[:C + 1], bucket[C + 1:]
        dWa, dba = b_att[:C].view(C, 1), b_att[C:]
        dWt, dbt = b_td[:C * K].view(C, K), b_td[C * K:]
        dX = torch.empty_like(X)
        ctr = torch.zeros(1, dtype=torch.int64, device=gpu)
        sync = None
        if overlapped:
            comms = [rccl.RcclCommunicator(0, 1, gpu), rccl.RcclCommunicator(0, 1, gpu)]
            sync = deploy.OverlappedGradientSum(b_att, b_td, comms[0], comms[1], gpu)

        def update_td():
            if overlapped:
                torch.cuda._sleep(3000000)          # >1 ms: far longer than a whole step
            Wt.add_(dWt, alpha=-lr)
            bt.add_(dbt, alpha=-lr)

        def update_att():
            Wa.add_(dWa, alpha=-lr)
            ba.add_(dba, alpha=-lr)
        losses = []
        hooks = sync.hooks if overlapped else None       # apa_hooks: passed per call, no library state
        for _ in range(4):
            logits, att, zsave, abar, _, _ = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, flags=flags, keep_prob=0.5,
                                                               seed=42, offset=ctr, workspace=ws, hooks=hooks)
            loss, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels)
            losses.append(loss.clone())
            cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, att, zsave, abar, G, flags=flags, keep_prob=0.5, seed=42,
                              offset=ctr, workspace=ws, out=(dX, None, dWa, dba, dWt, dbt), hooks=hooks)
            if overlapped:
                sync.after_backward(update_td, update_att)
            else:
                update_td()
                update_att()
        torch.cuda.synchronize()
        if overlapped:
            sync.close()
            for c in comms:
                c.close()
        return [t.cpu() for t in (Wa, ba, Wt, bt, dX, torch.stack(losses).flatten())]
    seq, ovl = run(False), run(True)
    for a, b, name in zip(seq, ovl, ('Wa', 'ba', 'Wt', 'bt', 'dX', 'losses')):
        assert torch.equal(a, b), name
    assert float(seq[5][0]) != float(seq[5][3 * (1 + N)])      # the weights did move


@pytest.mark.parametrize('N,K', [(1, 3), (5, 4), (32, 393), (33, 51), (64, 129), (65, 512), (200, 513),
                                 (7, 1024), (3, 1500)])
def test_softmax_xent_shapes_vs_float64(gpu, N, K):
    """Every code path of apa_softmax_xent_fwd_bwd (src/loss.py:74-80, eval.py:193-197): streaming
    fallback (K < 4, K > 1024), 1/2/4/8 vectors per lane, ragged last vector (K % 4 != 0), odd row
    pairs, loss-only single block, loss + gradient block roles (N <= 64), multi-block (N > 64)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    g = torch.Generator().manual_seed(N * 1000 + K)
    lg = torch.randn(N, K, generator=g) * 3
    lab = torch.randint(0, K, (N,), generator=g)
    ref_lp = torch.log_softmax(lg.double(), dim=1)
    ref_loss = -ref_lp[torch.arange(N), lab].mean()
    ref_G = (ref_lp.exp() - torch.nn.functional.one_hot(lab, K)) / N
    lb, G, probs, pred = cof.softmax_xent_fwd_bwd(lg.to(gpu), lab.to(gpu), want_probs=True, want_pred=True)
    assert abs(float(lb[0]) - float(ref_loss)) < 2e-6 * max(1.0, float(ref_loss))
    assert _rel(lb[1:].cpu().numpy(), -ref_lp[torch.arange(N), lab].numpy()) < 2e-6
    assert _rel(G.cpu().numpy(), ref_G.numpy()) < 5e-6
    assert _rel(probs.cpu().numpy(), ref_lp.exp().numpy()) < 5e-6
    assert torch.equal(pred.cpu(), lg.argmax(1))
    lb2, G2, _, _ = cof.softmax_xent_fwd_bwd(lg.to(gpu), lab.to(gpu), want_grad=False)   # loss only
    assert G2 is None and float(lb2[0]) == float(lb[0])


def test_device_label_path_bit_exact_vs_host_functions(gpu):
    """apa_pose_labels_device (one fused, canvas-free kernel per batch) == the host functions
    apa_pose_to_heatmap (no blur) -> *255 -> apa_pose_label_replay_resize, bit for bit: multiple
    people, missing keypoints (-1), keypoints on the border, crops, flips, 0.05 and 0.1 markers,
    plus the degenerate crops (nothing visible; a crop fully inside a disc)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    rng = np.random.RandomState(7)
    J = 16
    for ratio in (0.1, 0.05):
        poses, geoms = [], []
        for i in range(24):
            im_ht, im_wd = int(rng.randint(200, 700)), int(rng.randint(200, 700))
            n_people = int(rng.randint(1, 4))
            p = np.zeros((n_people, J, 3), dtype=np.int64)
            p[..., 0] = rng.randint(-20, im_wd + 20, size=(n_people, J))
            p[..., 1] = rng.randint(-20, im_ht + 20, size=(n_people, J))
            p[..., 2] = 1
            miss = rng.rand(n_people, J) < 0.3
            p[miss] = -1
            p[..., 0] = np.where(p[..., 0] >= im_wd, im_wd - 1, p[..., 0])
            p[..., 1] = np.where(p[..., 1] >= im_ht, im_ht - 1, p[..., 1])
            if i == 0:
                p[...] = -1                              # nothing visible -> all-zero label
            crop_h, crop_w = int(rng.randint(im_ht // 2, im_ht + 1)), int(rng.randint(im_wd // 2, im_wd + 1))
            crop_y, crop_x = int(rng.randint(0, im_ht - crop_h + 1)), int(rng.randint(0, im_wd - crop_w + 1))
            if i == 1:                                   # a tiny crop fully inside every joint's disc
                p[..., 0], p[..., 1], p[..., 2] = im_wd // 2, im_ht // 2, 1
                crop_h = crop_w = max(8, min(im_ht, im_wd) // 40)
                crop_y, crop_x = im_ht // 2 - crop_h // 2, im_wd // 2 - crop_w // 2
            # the crop is recorded on the image AFTER the aspect-preserving resize to RESIZE_SIDE
            # (vgg_preprocessing.py:325): a different frame from the keypoints' (original) one
            scale = 480.0 / min(im_ht, im_wd) if i % 3 else 1.0
            aug_ht, aug_wd = int(im_ht * scale), int(im_wd * scale)
            crop_y, crop_x = int(crop_y * scale), int(crop_x * scale)
            crop_h, crop_w = max(1, int(crop_h * scale)), max(1, int(crop_w * scale))
            if i == 1:
                aug_ht, aug_wd = im_ht, im_wd
                crop_h = crop_w = max(8, min(im_ht, im_wd) // 40)
                crop_y, crop_x = im_ht // 2 - crop_h // 2, im_wd // 2 - crop_w // 2
            poses.append(p.reshape(-1))
            geoms.append((im_ht, im_wd, aug_ht, aug_wd, crop_y, crop_x, crop_h, crop_w, int(rng.rand() < 0.5)))
        labels, valid, status = cof.pose_labels_device(poses, geoms, marker_wd_ratio=ratio, device=gpu)
        labels, valid, status = labels.cpu().numpy(), valid.cpu().numpy(), status.cpu().numpy()
        assert (status == 0).all()
        for i, (p, g) in enumerate(zip(poses, geoms)):
            hm, v = cof.pose_to_heatmap(p, g[0], g[1], 200, out_channels=J, marker_wd_ratio=ratio,
                                        do_gauss_blur=False)
            ref = cof.pose_label_replay_resize(hm, (g[2], g[3]), [g[4], g[5], g[6], g[7]], bool(g[8]), 15)
            assert np.array_equal(valid[i], v), i
            assert np.array_equal(labels[i], ref), (i, float(np.abs(labels[i] - ref).max()))
        assert labels[0].max() == 0.0 and labels[2:].max() == 1.0
    # a crop outside the image is reported, not silently accepted
    _, _, st = cof.pose_labels_device([poses[3]], [(300, 300, 300, 300, 250, 0, 100, 100, 0)], device=gpu)
    assert int(st[0]) == 1


def test_cfg001_baseline_head_matches_oracle(gpu):
    """BASELINE configs[0] (001_MPII_ResNet.yaml, no attention, eval batch 1): the shipped YAML loads
    unchanged and network_fn returns global-average-pool + logits (resnet_v1.py:206-217)."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    ref_yaml = '/root/reference/experiments/001_MPII_ResNet.yaml'
    cfg = apa_config.reset_cfg()
    if os.path.exists(ref_yaml):
        apa_config.cfg_from_file(ref_yaml)
    else:   # the GPU box has no reference tree: the same keys by hand
        apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'TEST': {'BATCH_SIZE': 1}})
    assert not cfg.NET.USE_POSE_PRELOGITS_BASED_ATTENTION
    fn = nets_factory.get_network_fn('resnet_v1_101', 393, 16, cfg, is_training=False, device=gpu)
    g = torch.Generator().manual_seed(3)
    X = torch.relu(torch.randn(1, 15, 15, 2048, generator=g))
    Xd = X.to(gpu).requires_grad_(True)
    logits, ep = fn(Xd)
    w, b = fn.head.logits_weights.detach().cpu().double(), fn.head.logits_biases.detach().cpu().double()
    ref = orc.baseline_avgpool_logits(X.double(), w, b)
    assert _rel(logits.detach().cpu().numpy(), ref.numpy()) < 2e-5 and 'Logits' in ep
    logits.sum().backward()
    assert _rel(Xd.grad.cpu().numpy(), np.broadcast_to((w.sum(1) / 225.0).numpy(), (1, 15, 15, 2048))) < 5e-5
    # training mode with the shipped YAML (NET.DROPOUT = -1): the backbone's dropout is NOT configured
    # (nets_factory.py:127-129 passes dropout_keep_prob only for DROPOUT >= 0; resnet_v1_101 defaults to 1.0)
    fnd = nets_factory.get_network_fn('resnet_v1_101', 393, 16, cfg, is_training=True, device=gpu)
    fnd.head.load_state_dict(fn.head.state_dict())
    assert fnd.head.keep_prob == 1.0
    assert _rel(fnd(X.to(gpu))[0].detach().cpu().numpy(), ref.numpy()) < 2e-5
    # NET.DROPOUT = 0.8: dropout on the pooled vector (keep 0.2), in HIP both ways -- the oracle is handed the
    # op's own mask (the counter-based stream over the N*C pooled elements, one offset per step)
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    apa_config.cfg_from_dict({'NET': {'DROPOUT': 0.8}})
    fnt = nets_factory.get_network_fn('resnet_v1_101', 393, 16, cfg, is_training=True, device=gpu)
    assert abs(fnt.head.keep_prob - 0.2) < 1e-12
    fnt.head.load_state_dict(fn.head.state_dict())
    fnt.head._step = 0
    X2 = torch.relu(torch.randn(3, 7, 7, 2048, generator=g))
    for step in range(2):
        Xt = X2.to(gpu).requires_grad_(True)
        lt, _ = fnt(Xt)
        mask = cof.dropout_mask((3, 1, 1, 2048), fnt.head.keep_prob, fnt.head.seed, step, device=gpu).cpu()
        assert 0.1 < float(mask.float().mean()) < 0.3
        Xr = X2.double().requires_grad_(True)
        ref_t = orc.baseline_avgpool_logits(Xr, w, b, True, fnt.head.keep_prob, mask.double())
        assert _rel(lt.detach().cpu().numpy(), ref_t.detach().numpy()) < 2e-5, step
        wsum = torch.randn(3, 393, generator=g)
        (lt * wsum.to(gpu)).sum().backward()
        (ref_t * wsum.double()).sum().backward()
        assert _rel(Xt.grad.cpu().numpy(), Xr.grad.numpy()) < 5e-5, step
    assert fnt.head.get_extra_state()['dropout_step'] == 2
    acc = sum(fnt(X.to(gpu))[0].detach() for _ in range(200)) / 200      # unbiased in expectation
    assert float((acc.cpu() - logits.detach().cpu()).abs().max()) < 0.25 * float(logits.detach().abs().max()) + 0.05
    # bf16 features: the backward writes the feature dtype
    Xb = X2.to(gpu).bfloat16().requires_grad_(True)
    fnt(Xb)[0].sum().backward()
    assert Xb.grad.dtype == torch.bfloat16 and bool(torch.isfinite(Xb.grad.float()).all())
    apa_config.reset_cfg()


def test_yaml_driven_training_loop_matches_cpu_reference_loop(gpu):
    """The pieces together the way src/train.py drives them: cfg 002 YAML -> network_fn -> gen_losses
    -> flat gradient bucket -> ITER_SIZE accumulation (averaged, train.py:547-566) -> fused
    momentum-SGD with L2 on the conv weights and the staircase learning rate.  Three parameter updates
    on a fixed synthetic batch must land where the same loop on the CPU oracle + torch.optim.SGD
    lands (dropout off: cfg.NET.DROPOUT = 0, so both loops are deterministic)."""
    from attentionalpoolingaction_amd import config as apa_config, deploy, loss as apa_loss, nets_factory
    ref_yaml = '/root/reference/experiments/002_MPII_ResNet_withAttention.yaml'
    cfg = apa_config.reset_cfg()
    if os.path.exists(ref_yaml):
        apa_config.cfg_from_file(ref_yaml)
    else:
        apa_config.cfg_from_dict({'MODEL_NAME': 'resnet_v1_101', 'TRAIN': {'ITER_SIZE': 2}, 'NET': {
            'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
            'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True}})
    apa_config.cfg_from_dict({'NET': {'DROPOUT': 0.0}})
    assert cfg.TRAIN.ITER_SIZE == 2 and apa_config.dropout_keep_prob(cfg) == 1.0
    K, N, wd, lr0 = 10, 8, cfg.TRAIN.WEIGHT_DECAY, 5e-3
    fn = nets_factory.get_network_fn(cfg.MODEL_NAME, K, 16, cfg, weight_decay=wd, is_training=True, device=gpu)
    head = fn.head
    g = torch.Generator().manual_seed(0)
    names = ['att_weights', 'att_biases', 'td_weights', 'td_biases']
    with torch.no_grad():
        head.att_weights.copy_(torch.randn(2048, 1, generator=g) / 45)
        head.td_weights.copy_(torch.randn(2048, K, generator=g) / 45)
    X = torch.relu(torch.randn(2, N, 14, 14, 2048, generator=g)) * 0.5      # two micro-batches
    y = torch.randint(0, K, (2, N), generator=g)
    ref = {k: getattr(head, k).detach().cpu().double().clone().requires_grad_(True) for k in names}
    opt_ref = torch.optim.SGD([{'params': [ref['att_weights'], ref['td_weights']], 'weight_decay': wd},
                               {'params': [ref['att_biases'], ref['td_biases']], 'weight_decay': 0.0}],
                              lr=lr0, momentum=0.9)
    params = {k: getattr(head, k) for k in names}
    bucket = deploy.GradientBucket({k: v.shape for k, v in params.items()}, gpu)
    accum = deploy.GradientAccumulator(bucket, cfg.TRAIN.ITER_SIZE)
    opt = deploy.MomentumSGD(params, bucket, lr=lr0, momentum=0.9, weight_decay=wd,
                             regularized=['att_weights', 'td_weights'])
    Xd, yd = X.to(gpu), y.to(gpu)
    for step in range(3):
        lr = deploy.exponential_decay_lr(lr0, step, 2, cfg.TRAIN.LEARNING_RATE_DECAY_RATE)
        for grp in opt_ref.param_groups:
            grp['lr'] = lr
        opt_ref.zero_grad()
        for mb in range(cfg.TRAIN.ITER_SIZE):
            logits, ep = fn(Xd[mb])
            total = sum(apa_loss.gen_losses(yd[mb], logits, cfg.TRAIN.LOSS_FN_ACTION, K, 1.0, None, None, '',
                                            None, 1.0, ep, cfg))
            grads = torch.autograd.grad(total, [params[k] for k in names])
            for k, gk in zip(names, grads):
                bucket.views[k].copy_(gk)
            if accum.step():
                opt.step(lr=lr)
            lr_, _ = orc.attentional_pooling(X[mb].double(), None, None, [ref['att_weights']], [ref['att_biases']],
                                             [ref['td_weights']], [ref['td_biases']], orc.AttnFlags())
            (orc.action_softmax_xent(lr_, y[mb], K) / cfg.TRAIN.ITER_SIZE).backward()   # averaged over micro-steps
        opt_ref.step()
    for k in names:
        assert _rel(params[k].detach().cpu().numpy(), ref[k].detach().numpy()) < 2e-5, k
    apa_config.reset_cfg()


@pytest.mark.parametrize('launcher', ['torchrun', 'self'])
def test_bench_two_ranks_sharing_one_gpu_over_gloo(gpu, launcher):
    """The driver's N > 1 launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py
    --gpus N`) on this single-GPU box: two ranks share the GPU and sum their gradient buckets over gloo
    (`--comm gloo`).  Covers rendezvous on 127.0.0.1, per-rank inputs, the barrier-bracketed timing with
    the max over ranks, whole-job throughput, rank-0-only JSON and the teardown -- everything of the
    N > 1 flow except the RCCL transport itself.  `self`: the plain `python bench.py --gpus 2 ...` command,
    which must start its own ranks and print the same single line."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    tail = [os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '3', '--comm', 'gloo']
    head = [sys.executable] if launcher == 'self' else [
        sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
        '--master-addr', '127.0.0.1', '--master-port', str(port)]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run(head + tail, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 64 and d['config']['parallelism'] == 'dp2'
    assert d['scaling'] == 'weak' and d['value'] > 0 and 'cpu_baseline' not in d
    assert abs(d['value'] - 64 / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']   # whole-job images / max-rank time
    assert 'gloo' in d['config']['comm']
    # what the communicator itself reports, and the measurement the overlap decision is taken from
    cd = d['config']['comm_detail']
    assert cd['ranks'] == 2 and cd['allreduce_bucket_bytes'] == (2048 + 1 + 2048 * 393 + 393) * 4
    assert cd['allreduce_us'] > 0 and cd['overlap'] in ('on', 'off')


def test_bench_prints_one_json_line_with_the_contract_fields(gpu):
    """bench.py's stdout contract: exactly one JSON line carrying the driver's fields plus the
    `roofline` and `cpu_baseline` objects; also through the N > 1 code path (1-rank RCCL group)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (['--cpu-seconds', '1'], ['--no-cpu-baseline', '--force-dist']):
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '20', '--warmup', '3'] + extra,
                             capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                  'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
            assert k in d, k
        assert d['unit'] == 'images/sec' and d['steps'] == 20 and d['warmup'] == 3 and d['n_gpus'] == 1
        assert d['vs_baseline'] is None and d['dtype'] == 'f32' and 'workload' in d['config']
        if '--force-dist' in extra:      # a one-rank RCCL group: ncclCommCount and the measured bucket all-reduce
            cd = d['config']['comm_detail']
            assert cd['ranks'] == 1 and cd['allreduce_us'] > 0
            pr = cd['overlap_probe_us']          # --overlap auto: both schedules timed at start-up, the faster kept
            assert pr['two_streams'] > 0 and pr['in_stream'] > 0
            assert cd['overlap'] == ('on' if pr['two_streams'] < pr['in_stream'] else 'off')
        else:
            assert d['config']['comm_detail'] is None
        r = d['roofline']
        assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
        assert 0.2 < r['frac'] < 1.0 and d['value'] > 2000          # BASELINE target: >= 2000 img/s
        assert d['repeats'] >= 5 and d['timed_ms_total'] >= 50.0       # >= 50 ms measured, median of >= 5 loops
        assert 'rotated over' in d['config']['workload'] and r['timer'].startswith('hipExtLaunchKernel') and (r['rocprofv3'] is None or 0.3 < r['rocprofv3']['frac'] < 1.0)
        assert d['roofline_fwd']['kernel'] == 'm1s_pool_fwd_kernel' and 0.2 < d['roofline_fwd']['frac'] < 1.0
        if '--no-cpu-baseline' not in extra:
            c = d['cpu_baseline']
            assert c['kind'] == 'port' and c['unit'] == 'images/sec' and c['cores'] >= 1 and c['value'] > 0
            # the other BASELINE configs ride on the same line
            for k in ('cfg002_eval', 'cfg003_bf16_train', 'hmdb51_perclass_bf16_train', 'hmdb51_rank1_bf16_train',
                      'cfg002_train_n512'):
                assert 'error' not in d['extra'][k], d['extra'][k]
                assert d['extra'][k]['ms_per_step'] > 0 and 0.0 < d['extra'][k]['roofline']['frac'] < 1.0
            # ... and BASELINE configs[1] / [2] end to end, with the torch-ROCm ResNet-101 in front of the HIP head
            for k in ('cfg002_eval_e2e', 'cfg003_train_e2e'):
                assert 'error' not in d['extra'][k], d['extra'][k]
                assert d['extra'][k]['images_per_sec'] > 100 and 0.0 < d['extra'][k]['head_share']['fraction_of_step'] < 0.2
            it = d['extra']['cfg002_train_iter_size']
            assert 'error' not in it, it
            assert it['iter_size'] == 2 and 0.2 < it['overlapped']['step_roofline_frac'] < 1.0
            assert it['overlapped']['us_per_update'] < 1.05 * it['sequential']['us_per_update']
            assert it['one_pass_batch_64']['us_per_update'] < it['sequential']['us_per_update']


@pytest.mark.parametrize('lanes,dtype', [(2, torch.float32), (3, torch.bfloat16)])
def test_overlapped_micro_batches_equal_the_sequential_gradient_accumulator(gpu, lanes, dtype):
    """TRAIN.ITER_SIZE (src/train.py:529-566): the micro-batches of one update on separate streams
    (deploy.OverlappedMicroBatches) give BIT-IDENTICAL accumulated gradients, dX and losses to the same micro-batches
    run one after the other through deploy.GradientAccumulator."""
    from attentionalpoolingaction_amd import deploy
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, C, K = 8, 49, 2048, 51
    g = torch.Generator().manual_seed(5)
    Wa, ba = (torch.randn(C, 1, generator=g) / 45).to(gpu), torch.zeros(1, device=gpu)
    Wt, bt = (torch.randn(C, K, generator=g) / 45).to(gpu), (torch.randn(K, generator=g) * 0.1).to(gpu)
    Xs = [torch.relu(torch.randn(N, P, C, generator=g)).to(gpu).to(dtype) for _ in range(lanes)]
    labels = [torch.randint(0, K, (N,), generator=g).to(gpu) for _ in range(lanes)]
    flags = cof.attn_flags(False, False, True)
    sizes = [C, 1, C * K, K]

    def views(flat):
        o, v = 0, []
        for n in sizes:
            v.append(flat[o:o + n])
            o += n
        return v[0].view(C, 1), v[1], v[2].view(C, K), v[3]

    def build(lane_buckets, dXs, ctrs):
        st = []
        for l in range(lanes):
            dWa, dba, dWt, dbt = views(lane_buckets[l])
            st.append(cof.HeadTrainStep(Xs[l], Xs[l], Wa, ba, Wt, bt, labels[l], (dXs[l], None, dWa, dba, dWt, dbt),
                                        flags=flags, keep_prob=0.2, seed=42 + l, offset=ctrs[l]))
        return st

    def fresh():
        return ([torch.zeros(sum(sizes), device=gpu) for _ in range(lanes)], [torch.empty_like(x) for x in Xs],
                [torch.zeros(1, dtype=torch.int64, device=gpu) for _ in range(lanes)])

    # (a) the overlapped schedule, three updates in a row (the dropout counters advance)
    lb, dXa, ctr = fresh()
    out_a = torch.zeros(sum(sizes), device=gpu)
    sched = deploy.OverlappedMicroBatches(build(lb, dXa, ctr), lb, out_a, gpu)
    hist_a = []
    for _ in range(3):
        sched.run()
        torch.cuda.synchronize()
        hist_a.append((out_a.clone(), [d.clone() for d in dXa], [s.loss.clone() for s in sched.steppers]))
    sched.close()
    assert [int(c) for c in ctr] == [3] * lanes
    # (b) the same micro-batches one at a time, accumulated by deploy.GradientAccumulator
    lb2, dXb, ctr2 = fresh()
    work = torch.zeros(sum(sizes), device=gpu)
    bucket = deploy.GradientBucket({'flat': (sum(sizes),)}, gpu)
    acc = deploy.GradientAccumulator(bucket, lanes)
    st2 = build([work] * lanes, dXb, ctr2)          # every micro-step writes the ONE working bucket
    for u in range(3):
        ready = False
        for l in range(lanes):
            st2[l].run()
            torch.cuda.synchronize()
            bucket.flat.copy_(work)
            ready = acc.step()
        assert ready
        got_flat, got_dX, got_loss = hist_a[u]
        assert torch.equal(bucket.flat, got_flat)
        for l in range(lanes):
            assert torch.equal(dXb[l], got_dX[l]) and torch.equal(st2[l].loss, got_loss[l])
    assert float(hist_a[0][0].abs().max()) > 0 and not torch.equal(hist_a[0][0], hist_a[1][0])


def test_accumulate_gradients_divides_like_the_reference_and_folds_more_than_eight_parts(gpu):
    """`ref_grad / float(ITER_SIZE)` (src/train.py:560-563) is a DIVISION -- bit-identical to
    torch's `.div_` for ITER_SIZE = 3, 5, 7 (where * (1/ITER_SIZE) is an ulp off on some elements) -- and an
    ITER_SIZE above APA_ACC_MAX_PARTS folds in groups with the sequential summation order."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    g = torch.Generator().manual_seed(11)
    n = 4099                                                   # vector body + scalar tail
    for k in (2, 3, 5, 7, 8, 9, 11, 17):
        parts = [torch.randn(n, generator=g).to(gpu) for _ in range(k)]
        ref = torch.zeros(n, device=gpu)
        for p_ in parts:
            ref.add_(p_)
        out = torch.empty(n, device=gpu)
        cof.accumulate_gradients(out, parts, divisor=float(k))
        # IEEE division (numpy on the host); NOT torch's `gpu_tensor / python_scalar`, which multiplies by 1/k
        want = torch.from_numpy(ref.cpu().numpy() / np.float32(k)).to(gpu)
        assert torch.equal(out, want), k
        assert torch.equal(out, torch.div(ref, torch.full((), float(k), device=gpu))), k
        cof.accumulate_gradients(out, parts, scale=0.25)
        assert torch.equal(out, ref * 0.25), k
    with pytest.raises(cof.ApaError):
        cof.accumulate_gradients(out, parts, divisor=0.0)
    with pytest.raises(cof.ApaError):
        cof.accumulate_gradients(out, parts)


def test_module_surface_hands_apa_hooks_to_the_kernels_and_pose_feat_with_13_keypoints(gpu):
    """(a) `head.hooks` reaches the forward AND the backward pooling call made by autograd (the grad-ready event is
    recorded by the library during backward); results are unaffected.  (b) _WITH_POSE_FEAT + per-class maps with
    J = 13 keypoints: C + J is not a whole number of 16-byte vectors, the literal concat is zero-padded."""
    from attentionalpoolingaction_amd import config as apa_config, nets_factory
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True}})
    fn = nets_factory.get_network_fn('resnet_v1_101', 51, 16, cfg, is_training=True, device=gpu, in_channels=256)
    g = torch.Generator().manual_seed(3)
    X = torch.relu(torch.randn(4, 5, 5, 256, generator=g)).to(gpu)
    labels = torch.randint(0, 51, (4,), generator=g).to(gpu)

    def run(hooks):
        fn.head._step = 0
        fn.head.hooks = hooks
        for p_ in fn.head.parameters():
            p_.grad = None
        Xd = X.clone().requires_grad_(True)
        logits, _ = fn(Xd)
        torch.nn.functional.cross_entropy(logits, labels).backward()
        return logits.detach().clone(), Xd.grad.clone(), fn.head.td_weights.grad.clone()
    base = run(None)
    t0, ready, tdw = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tdw.record()
    ready.record()
    torch.cuda.synchronize()
    t0.record()
    hooked = run(cof.make_hooks(grad_ready=ready, td_weights_ready=tdw))
    torch.cuda.synchronize()
    assert t0.elapsed_time(ready) > 0.0          # re-recorded by the backward call, after t0
    for a, b in zip(base, hooked):
        assert torch.equal(a, b)
    apa_config.reset_cfg()
    # (b)
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS': True,
                                      'USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT': True}})
    J, K, C = 13, 20, 64
    fn = nets_factory.get_network_fn('resnet_v1_101', K, J, cfg, is_training=False, device=gpu, in_channels=C)
    head = fn.head
    with torch.no_grad():
        for p_ in head.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) / max(p_.shape[0], 1) ** 0.5 if p_.dim() >= 2
                     else torch.randn(p_.shape, generator=g) * 0.1)
    X = torch.relu(torch.randn(2, 3, 4, C, generator=g))
    Xd = X.to(gpu).requires_grad_(True)
    logits, ep = fn(Xd)
    logits.sum().backward()
    p = {k: v.detach().cpu().double() for k, v in head.named_parameters()}
    Xr = X.double().requires_grad_(True)
    pre, pl = orc.pose_logits_head(Xr, p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'])
    lr, _ = orc.attentional_pooling(Xr, pre, pl, [p['att_weights']], [p['att_biases']], [p['td_weights']],
                                    [p['td_biases']], orc.AttnFlags(per_class=True, with_pose_feat=True))
    lr.sum().backward()
    assert _rel(logits.detach().cpu().numpy(), lr.detach().numpy()) < 5e-5
    assert _rel(Xd.grad.cpu().numpy(), Xr.grad.numpy()) < 1e-4
    apa_config.reset_cfg()
