"""CPU tests of the oracle: golden fixtures, internal consistency, and the algebra the HIP kernels
rely on (factorised forward / closed-form backward == the literal reference formulation)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import attn_pool_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _t(a, dtype=torch.float64):
    return torch.from_numpy(np.asarray(a)).to(dtype)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'attn_*.npz'))),
                         ids=lambda p: 'regression_pin_' + os.path.basename(p)[5:-4])
def test_oracle_matches_golden(path):
    d = np.load(path)
    fused = 'Xatt' not in d.files
    X = _t(d['X']).requires_grad_(True)
    Xatt = X if fused else _t(d['Xatt']).requires_grad_(True)
    Wa, ba, Wt, bt = (_t(d[k]).requires_grad_(True) for k in ('Wa', 'ba', 'Wt', 'bt'))
    flags = orc.AttnFlags(single_layer_att=fused, softmax_att=bool(d['softmax']), relu_att=bool(d['relu']))
    mask = torch.from_numpy(d['mask']) if bool(d['train']) else None
    logits, ep = orc.attentional_pooling(X, None if fused else Xatt, None, [Wa], [ba], [Wt], [bt],
                                         flags, is_training=bool(d['train']),
                                         keep_prob=float(d['keep']), dropout_mask=mask)
    loss = orc.action_softmax_xent(logits, torch.from_numpy(d['labels']), Wt.shape[1])
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), d['logits'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ep['PosePrelogitsBasedAttention'].detach().numpy(), d['att'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(float(loss), float(d['loss']), rtol=1e-12)
    np.testing.assert_allclose(X.grad.numpy(), d['dX'], rtol=2e-6, atol=1e-12)   # stored as f32
    np.testing.assert_allclose(Wt.grad.numpy(), d['dWt'], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(Wa.grad.numpy(), d['dWa'], rtol=1e-10, atol=1e-14)


def _rand_case(N=3, H=4, W=5, C=32, K=7, Ca=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    Ca = C if Ca is None else Ca
    X = torch.randn(N, H, W, C, generator=g, dtype=torch.float64)   # not relu'd: vgg tap can be < 0
    Xa = X if Ca == C else torch.randn(N, H, W, Ca, generator=g, dtype=torch.float64)
    Wa = torch.randn(Ca, 1, generator=g, dtype=torch.float64) * 0.3
    ba = torch.randn(1, generator=g, dtype=torch.float64) * 0.1
    Wt = torch.randn(C, K, generator=g, dtype=torch.float64) * 0.3
    bt = torch.randn(K, generator=g, dtype=torch.float64) * 0.1
    G = torch.randn(N, K, generator=g, dtype=torch.float64)
    return X, Xa, Wa, ba, Wt, bt, G


@pytest.mark.parametrize('softmax,relu,train', [(False, False, False), (True, False, False),
                                                (False, True, False), (False, False, True),
                                                (True, False, True)])
def test_factorised_forward_and_closed_form_backward(softmax, relu, train):
    """The M == 1 algebra in apa_m1.hip's header, checked against autograd of the literal graph:
         z = (1/P) sum_p A Xt ; logits = z Wt + abar bt
         dz = G Wt^T ; dWt = z^T G ; dbt = sum_n abar G
         dA = (Xt.dz + G.bt)/P ; dZ = dA | dA*[A>0] | A*(dA - (z.dz + (G.bt) abar))
         dX = (A/P) dz * mask/keep + dZ wa ; dwa = sum dZ X ; dba = sum dZ"""
    X, _, Wa, ba, Wt, bt, G = _rand_case(seed=3)
    N, H, W, C = X.shape
    P = H * W
    keep = 0.4
    g = torch.Generator().manual_seed(9)
    mask = (torch.rand(X.shape, generator=g) < keep) if train else None
    Xl = X.clone().requires_grad_(True)
    Wal, bal, Wtl, btl = (t.clone().requires_grad_(True) for t in (Wa, ba, Wt, bt))
    flags = orc.AttnFlags(softmax_att=softmax, relu_att=relu)
    logits, ep = orc.attentional_pooling(Xl, None, None, [Wal], [bal], [Wtl], [btl], flags,
                                         is_training=train, keep_prob=keep, dropout_mask=mask)
    (logits * G).sum().backward()

    A = ep['PosePrelogitsBasedAttention'].detach().reshape(N, P)
    Xf = X.reshape(N, P, C)
    mk = (mask.reshape(N, P, C).double() / keep) if train else torch.ones_like(Xf)
    Xt = Xf * mk
    z = (A[:, :, None] * Xt).sum(1) / P
    abar = A.sum(1) / P
    np.testing.assert_allclose((z @ Wt + abar[:, None] * bt).numpy(), logits.detach().numpy(), rtol=1e-10, atol=1e-12)
    dz = G @ Wt.t()
    sn = G @ bt
    dA = ((Xt * dz[:, None, :]).sum(-1) + sn[:, None]) / P
    if softmax:
        corr = (z * dz).sum(-1) + sn * abar
        dZ = A * (dA - corr[:, None])
    elif relu:
        dZ = dA * (A > 0)
    else:
        dZ = dA
    dX = (A / P)[:, :, None] * dz[:, None, :] * mk + dZ[:, :, None] * Wa[:, 0]
    np.testing.assert_allclose(dX.reshape(X.shape).numpy(), Xl.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose((z.t() @ G).numpy(), Wtl.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose((abar[:, None] * G).sum(0).numpy(), btl.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose((dZ[:, :, None] * Xf).sum((0, 1)).numpy(), Wal.grad[:, 0].numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(float(dZ.sum()), float(bal.grad), rtol=1e-8, atol=1e-11)


def test_per_class_and_rank_shapes():
    """_PER_CLASS gives K bottom-up maps (nets_factory.py:257); rank > 1 CHAINS the attention
    convs (`net` is re-assigned at :263), stacks on a new last axis and sums it out (:326-328)."""
    X, _, _, _, Wt, bt, _ = _rand_case(K=5)
    N, H, W, C = X.shape
    g = torch.Generator().manual_seed(1)
    Wa_k = torch.randn(C, 5, generator=g, dtype=torch.float64)
    ba_k = torch.zeros(5, dtype=torch.float64)
    logits, ep = orc.attentional_pooling(X, None, None, [Wa_k], [ba_k], [Wt], [bt],
                                         orc.AttnFlags(per_class=True))
    assert ep['PosePrelogitsBasedAttention'].shape == (N, H, W, 5) and logits.shape == (N, 5)
    want = ((X @ Wa_k + ba_k) * (X @ Wt + bt)).mean((1, 2))
    np.testing.assert_allclose(logits.numpy(), want.numpy(), rtol=1e-12)
    # rank 2, class-agnostic: second attention conv consumes the [.,1] output of the first
    Wa1 = torch.randn(C, 1, generator=g, dtype=torch.float64)
    Wa2 = torch.randn(1, 1, generator=g, dtype=torch.float64)
    Wt2 = torch.randn(C, 5, generator=g, dtype=torch.float64)
    z1 = X @ Wa1
    z2 = z1 @ Wa2
    logits2, ep2 = orc.attentional_pooling(X, None, None, [Wa1, Wa2], [torch.zeros(1).double()] * 2,
                                           [Wt, Wt2], [bt, bt], orc.AttnFlags(rank=2))
    assert ep2['PosePrelogitsBasedAttention'].shape == (N, H, W, 1, 2)
    want2 = (z1 * (X @ Wt + bt)).mean((1, 2)) + (z2 * (X @ Wt2 + bt)).mean((1, 2))
    np.testing.assert_allclose(logits2.numpy(), want2.numpy(), rtol=1e-12)


def test_pose_l2_loss_literal_scale_and_golden():
    d = np.load(os.path.join(GOLD, 'losses.npz'))
    Pl = _t(d['pose_Pl']).requires_grad_(True)
    lbl, valid = _t(d['pose_lbl']), torch.from_numpy(d['pose_valid'])
    loss = orc.pose_l2_loss(Pl, lbl, valid, float(d['pose_wt']))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(d['pose_loss']), rtol=1e-12)
    np.testing.assert_allclose(Pl.grad.numpy(), d['pose_dPl'], rtol=1e-12, atol=1e-16)
    # effective scale 0.5 / (N^2 * H * W): the reduce_sum(mask) divisor includes the batch (loss.py:54-56)
    N, H, W, J = Pl.shape
    direct = 0.5 * (((Pl.detach() - lbl) ** 2).sum((1, 2)) * valid.double()).sum() / (N * N * H * W)
    np.testing.assert_allclose(float(loss), float(direct) * float(d['pose_wt']), rtol=1e-12)
    # label maps of a different size are resized with the TF1 legacy bilinear rule first
    big = torch.rand(N, 2 * H, 2 * W, J, dtype=torch.float64)
    l2 = orc.pose_l2_loss(Pl.detach(), big, valid)
    np.testing.assert_allclose(float(l2), float(orc.pose_l2_loss(Pl.detach(), big[:, ::2, ::2], valid)), rtol=1e-12)


def test_softmax_xent_matches_torch_and_golden():
    d = np.load(os.path.join(GOLD, 'losses.npz'))
    logits = _t(d['xent_logits']).requires_grad_(True)
    labels = torch.from_numpy(d['xent_labels'])
    loss = orc.action_softmax_xent(logits, labels, logits.shape[1], float(d['xent_wt']))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(d['xent_loss']), rtol=1e-12)
    np.testing.assert_allclose(logits.grad.numpy(), d['xent_G'], rtol=1e-12, atol=1e-16)
    ref = torch.nn.functional.cross_entropy(logits.detach(), labels) * float(d['xent_wt'])
    np.testing.assert_allclose(float(loss), float(ref), rtol=1e-12)


def test_tf1_legacy_bilinear_resize():
    d = np.load(os.path.join(GOLD, 'losses.npz'))
    out = orc.tf1_resize_bilinear(_t(d['resize_in']), 4, 5)
    np.testing.assert_allclose(out.numpy(), d['resize_out'], rtol=1e-13)
    x = torch.arange(16, dtype=torch.float64).reshape(1, 4, 4, 1)
    # integer scale 2: src = dst*2 exactly -> plain sub-sampling of the top-left phase (no half pixel)
    np.testing.assert_allclose(orc.tf1_resize_bilinear(x, 2, 2)[0, :, :, 0].numpy(), [[0, 2], [8, 10]])
    # up-sampling x2: src = dst/2, last sample clamps (hi = min(lo+1, in-1))
    up = orc.tf1_resize_bilinear(x[:, :1, :2], 1, 4)[0, 0, :, 0].numpy()
    np.testing.assert_allclose(up, [0.0, 0.5, 1.0, 1.0])


def test_frame_pooling_and_temporal_attention():
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(6, 4, generator=g, dtype=torch.float64)     # B=2 videos x F=3 frames
    pooled, ep = orc.frame_pooling(logits, 3)
    np.testing.assert_allclose(pooled.numpy(), logits.reshape(2, 3, 4).mean(1).numpy())
    assert ep['logits_beforePool'] is logits
    w = torch.randn(4, 1, generator=g, dtype=torch.float64) * 0.001
    b = torch.full((1,), 1.0 / 3, dtype=torch.float64)                # bias init 1/F (:366-368)
    pooled2, ep2 = orc.frame_pooling(logits, 3, w, b)
    x = logits.reshape(2, 3, 4)
    np.testing.assert_allclose(pooled2.numpy(), (x * (x @ w + b)).mean(1).numpy())


def test_dp_clone_semantics():
    """model_deploy.py: loss/num_clones per tower + add_n of tower grads == gradient of the mean."""
    X, _, Wa, ba, Wt, bt, _ = _rand_case(N=4, seed=5)
    labels = torch.tensor([1, 3, 0, 2])
    Wt_full = Wt.clone().requires_grad_(True)
    lg, _ = orc.attentional_pooling(X, None, None, [Wa], [ba], [Wt_full], [bt], orc.AttnFlags())
    orc.action_softmax_xent(lg, labels, Wt.shape[1]).backward()
    grads = []
    for half in (slice(0, 2), slice(2, 4)):
        w = Wt.clone().requires_grad_(True)
        lg, _ = orc.attentional_pooling(X[half], None, None, [Wa], [ba], [w], [bt], orc.AttnFlags())
        orc.dp_clone_loss([orc.action_softmax_xent(lg, labels[half], Wt.shape[1])], 2).backward()
        grads.append([w.grad])
    np.testing.assert_allclose(orc.dp_sum_clone_grads(grads)[0].numpy(), Wt_full.grad.numpy(), rtol=1e-10, atol=1e-14)
