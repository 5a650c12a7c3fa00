"""The reference's training loop (model_deploy.py clones + src/train.py's accumulation / schedule / optimiser,
tests/golden/ref_train_*.npz) replayed with the PRODUCT end to end on the GPU: the HIP head computes every
clone's gradient (with the reference's dropout masks replayed), deploy.* sums / accumulates them and the fused
HIP momentum-SGD launch applies the update.  fp32 against the reference's float64."""
import os

import numpy as np
import pytest
import torch

import _ref_fixture as rf
from test_train_reference_cpu import TRAIN_PATHS, _id, replay

pytestmark = pytest.mark.gpu


def _rel(got, exp, floor=1e-30):
    return float(np.abs(np.asarray(got, dtype=np.float64) - exp).max() / max(np.abs(exp).max(), floor))


@pytest.mark.parametrize('path', TRAIN_PATHS, ids=_id)
def test_hip_training_loop_matches_reference(gpu, path):
    from attentionalpoolingaction_amd import loss as apa_loss
    tf_ = rf.TrainFixture(path)
    m = tf_.meta
    wd = m['weight_decay']
    built = {}

    def clone_gradients(params, b, d):
        fx = tf_.clone_fixture({vn: p.detach().cpu().numpy() for vn, p in params.items()}, b, d)
        if not built:                                  # one module for the whole loop: `params` ARE its tensors
            network_fn, cfg = rf.build_head(fx, device=gpu)
            table = rf.module_tf_names(network_fn)
            assert set(table) == set(params), sorted(set(table) ^ set(params))
            with torch.no_grad():
                for vn, t in table.items():
                    t.copy_(params[vn].reshape(t.shape))
                    params[vn] = t                      # the optimiser now updates the module's own tensors
            built.update(network_fn=network_fn, cfg=cfg, table=table)
        network_fn, cfg, table = built['network_fn'], built['cfg'], built['table']
        images = torch.from_numpy(fx.arrays['in/images']).to(gpu)
        network_fn.head.replay_dropout_mask(torch.from_numpy(fx.dropout_mask()).to(gpu))
        logits, ep = network_fn(images)
        tc = m['train_cfg']
        use_pose = bool(tc['LOSS_FN_POSE'])
        losses = apa_loss.gen_losses(
            torch.from_numpy(fx.arrays['in/labels_action']).to(gpu), logits, tc['LOSS_FN_ACTION'], m['num_classes'],
            tc['LOSS_FN_ACTION_WT'],
            torch.from_numpy(fx.arrays['in/labels_pose']).to(gpu) if use_pose else None,
            ep.get('PoseLogits') if use_pose else None, tc['LOSS_FN_POSE'] if use_pose else '',
            torch.from_numpy(fx.arrays['in/labels_pose_valid']).to(gpu) if use_pose else None, tc['LOSS_FN_POSE_WT'],
            ep, cfg)
        ts = [table[vn] for vn in m['var_order']]
        gs = torch.autograd.grad(sum(losses), ts, allow_unused=True)
        grads = {vn: (torch.zeros_like(t) if g is None else g) for vn, t, g in zip(m['var_order'], ts, gs)}
        return grads, [float(l.detach()) for l in losses]

    def on_run(r, run, params, bucket, clone_losses):
        for vn in m['grad_vars']:
            if vn in m['reg_only_grad']:
                assert float(bucket.views[vn].abs().max()) == 0.0
                continue
            exp = tf_.arrays['run/%d/grad/%s' % (r, vn)]
            got = bucket.views[vn].cpu().numpy().astype(np.float64)
            if vn.endswith('/weights'):
                got = got + wd * params[vn].detach().cpu().numpy().astype(np.float64).reshape(got.shape)
            assert _rel(got, exp.reshape(got.shape)) < 5e-5, (r, vn)
        for ls, exp in zip(clone_losses, run['clone_losses']):
            assert np.allclose(ls, exp, rtol=2e-5, atol=0)

    history, opt = replay(tf_, clone_gradients, dtype=torch.float32, device=gpu, on_run=on_run)
    assert opt.bucket.flat.is_cuda                      # the fused HIP update ran, not the torch expressions
    for s, vars_ in enumerate(history):
        for vn, got in vars_.items():
            key = 'step/%d/var/%s' % (s, vn)
            if key in tf_.arrays:
                exp = tf_.arrays[key]
                assert _rel(got.reshape(exp.shape), exp) < 2e-5, key
    # the step moved the weights by far more than the tolerance (the comparison is not vacuous)
    last = len(history) - 1
    vn = 'PosePrelogitsBasedAttention/Conv/weights'
    moved = _rel(tf_.arrays['var0/' + vn].astype(np.float64), tf_.arrays['step/%d/var/%s' % (last, vn)])
    assert moved > 5e-3
