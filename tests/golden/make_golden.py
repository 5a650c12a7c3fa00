#!/usr/bin/env python
"""Regenerates tests/golden/*.npz from the CPU oracle (float64).

    python tests/golden/make_golden.py

The reference itself (TF 1.1 / OpenCV) cannot run in this image and ships no golden vectors for
this path, so these fixtures are frozen outputs of the RESTATEMENT in oracle/ --
they pin the oracle against regressions and give the GPU tests a travelling target; they do not
pin the reference ("parity unpinned").  Small shapes only (C = 256, the smallest channel count
the HIP kernels are built for) so the fixtures stay < 1 MB.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import attn_pool_oracle as orc          # noqa: E402
from oracle import labels_eval_oracle as leo        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

ATTN_CASES = {
    # name: (N, H, W, C, K, Ca, softmax, relu, train)
    'cfg002_id': (3, 7, 7, 256, 393, 256, False, False, False),
    'softmax': (2, 5, 6, 256, 51, 256, True, False, False),
    'relu': (2, 4, 4, 256, 10, 256, False, True, False),
    'train_dropout': (2, 5, 5, 256, 51, 256, False, False, True),
    'cfg003_sep_att': (2, 5, 5, 256, 51, 96, False, False, False),
    'cfg003_sep_att_softmax': (2, 4, 5, 256, 20, 96, True, False, False),
}


def attn_case(name, spec):
    N, H, W, C, K, Ca, softmax, relu, train = spec
    g = torch.Generator().manual_seed(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    X = torch.relu(torch.randn(N, H, W, C, generator=g)).float()
    fused = Ca == C
    Xatt = X if fused else torch.relu(torch.randn(N, H, W, Ca, generator=g)).float()
    Wa = (torch.randn(Ca, 1, generator=g) / Ca ** 0.5).float()
    ba = (torch.randn(1, generator=g) * 0.1).float()
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).float()
    bt = (torch.randn(K, generator=g) * 0.1).float()
    labels = torch.randint(0, K, (N,), generator=g)
    keep = 0.5
    mask = (torch.rand(N, H, W, C, generator=g) < keep) if train else None

    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xd, Wad, bad, Wtd, btd = leaf(X), leaf(Wa), leaf(ba), leaf(Wt), leaf(bt)
    Xattd = Xd if fused else leaf(Xatt)
    flags = orc.AttnFlags(single_layer_att=fused, softmax_att=softmax, relu_att=relu)
    logits, ep = orc.attentional_pooling(Xd, None if fused else Xattd, None, [Wad], [bad], [Wtd],
                                         [btd], flags, is_training=train, keep_prob=keep,
                                         dropout_mask=mask)
    loss = orc.action_softmax_xent(logits, labels, K, 1.0)
    loss.backward()
    out = dict(X=X.numpy(), Wa=Wa.numpy(), ba=ba.numpy(), Wt=Wt.numpy(), bt=bt.numpy(),
               labels=labels.numpy(), softmax=softmax, relu=relu, train=train, keep=keep,
               logits=logits.detach().numpy(), att=ep['PosePrelogitsBasedAttention'].detach().numpy(),
               loss=loss.detach().numpy(), dX=Xd.grad.numpy().astype(np.float32),
               dWa=Wad.grad.numpy(), dba=bad.grad.numpy(),
               dWt=Wtd.grad.numpy().astype(np.float32), dbt=btd.grad.numpy())
    if K <= 64:     # the [N,H,W,K] end point is only kept for the small-K cases
        out['topdown'] = ep['TopDownAttention'].detach().numpy().astype(np.float32)
    if not fused:
        out['Xatt'] = Xatt.numpy()
        out['dXatt'] = Xattd.grad.numpy().astype(np.float32)
    if train:
        out['mask'] = mask.numpy().astype(np.uint8)
    return out


def loss_cases():
    g = torch.Generator().manual_seed(7)
    N, H, J, K = 3, 5, 16, 11
    Pl = torch.randn(N, H, H, J, generator=g).double().requires_grad_(True)
    lbl = torch.rand(N, H, H, J, generator=g).double()
    valid = torch.rand(N, J, generator=g) > 0.3
    loss = orc.pose_l2_loss(Pl, lbl, valid, 0.7)
    loss.backward()
    logits = torch.randn(N, K, generator=g).double().requires_grad_(True)
    labels = torch.randint(0, K, (N,), generator=g)
    xl = orc.action_softmax_xent(logits, labels, K, 1.3)
    xl.backward()
    lbl_big = torch.rand(2, 9, 7, 3, generator=g).double()
    return dict(pose_Pl=Pl.detach().numpy(), pose_lbl=lbl.numpy(), pose_valid=valid.numpy(),
                pose_wt=0.7, pose_loss=loss.detach().numpy(), pose_dPl=Pl.grad.numpy(),
                xent_logits=logits.detach().numpy(), xent_labels=labels.numpy(), xent_wt=1.3,
                xent_loss=xl.detach().numpy(), xent_G=logits.grad.numpy(),
                resize_in=lbl_big.numpy(), resize_out=orc.tf1_resize_bilinear(lbl_big, 4, 5).numpy())


def label_eval_cases():
    rng = np.random.RandomState(3)
    # the one input the reference supplies: src/custom_ops/test/pose_to_heatmap_op_test.py:10-23
    pose = [50, 50, 1] * 3 + [0, 0, 1] * 2 + [-1, -1, 1] * 11
    pose += [90, 90, 1] * 3 + [0, 0, 1] * 2 + [-1, -1, 1] * 11
    hm_ref, valid_ref = leo.pose_to_heatmap(pose, 100, 200, 100, out_channels=16)
    hm_nb, _ = leo.pose_to_heatmap(pose, 100, 200, 100, out_channels=16, do_gauss_blur=False)
    # training call geometry: out_wd = 200, ratio 0.05, no blur (preprocess_pipeline.py:155-163)
    pose2 = rng.randint(-1, 480, size=(2 * 16 * 3,)).tolist()
    hm_tr, valid_tr = leo.pose_to_heatmap(pose2, 360, 480, 200, out_channels=16,
                                          marker_wd_ratio=0.05, do_gauss_blur=False)
    scores = rng.rand(40, 7).astype(np.float32)
    scores[5] = scores[6]                       # exact ties
    labels = rng.randint(0, 6, size=(40,))      # class 6 never positive -> skipped
    mAP, aps = leo.compute_map(scores, labels)
    return dict(ref_pose=np.array(pose, dtype=np.int64), ref_hm=hm_ref, ref_hm_noblur=hm_nb,
                ref_valid=valid_ref, train_pose=np.array(pose2, dtype=np.int64),
                train_hm_packed=np.packbits(hm_tr > 0), train_hm_shape=np.array(hm_tr.shape),
                train_valid=valid_tr, map_scores=scores, map_labels=labels, map_value=mAP,
                map_aps=np.array(aps))


def main():
    for name, spec in ATTN_CASES.items():
        np.savez_compressed(os.path.join(HERE, 'attn_{}.npz'.format(name)), **attn_case(name, spec))
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **loss_cases())
    np.savez_compressed(os.path.join(HERE, 'labels_eval.npz'), **label_eval_cases())
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith('.npz'))
    print('wrote golden fixtures, {:.1f} KB total'.format(tot / 1024))


if __name__ == '__main__':
    main()
