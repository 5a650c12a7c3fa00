"""GPU tests of the STATE deploy.FusedHeadStep and its optimiser carry between steps (round 6; VERDICT r05 Weak #1,
ADVICE r05): operand images that live in a bound step's workspace, bf16 shadows, the per-shape step cache and the
staleness guard.  The yardstick is always a twin head driven through the stateless arm (no images, no shadow: every
call prepares its operands from the fp32 weights) -- results must be BIT-identical."""
import pytest
import torch

from attentionalpoolingaction_amd import config as apa_config, deploy, nets_factory

pytestmark = pytest.mark.gpu

PER_CLASS = {'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                     'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True,
                     'USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS': True},
             'TRAIN': {'LOSS_FN_POSE': ''}}       # the 002 yaml clears it; src/config.py:116 defaults to 'l2'
CFG003 = {'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True},
          'TRAIN': {'LOSS_FN_POSE': 'l2', 'LOSS_FN_POSE_WT': 1.0}}


def _twin_heads(gpu, table, K, seed=5):
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict(table)
    fns = [nets_factory.get_network_fn('resnet_v1_101', K, 16, cfg, weight_decay=5e-4, is_training=True, device=gpu)
           for _ in range(2)]
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for (n, p), (_, q) in zip(fns[0].head.named_parameters(), fns[1].head.named_parameters()):
            w = torch.randn(p.shape, generator=g) / (p.shape[0] ** 0.5 if p.dim() == 2 else 10.0)
            p.copy_(w.to(gpu))
            q.copy_(w.to(gpu))
    return cfg, fns


def _batch(gpu, N, K, seed, pose=False):
    g = torch.Generator().manual_seed(seed)
    X = torch.relu(torch.randn(N, 7, 7, 2048, generator=g)).to(torch.bfloat16).to(gpu)
    y = torch.randint(0, K, (N,), generator=g).to(gpu)
    if not pose:
        return X, y, None, None
    return X, y, torch.rand(N, 7, 7, 16, generator=g).to(gpu), (torch.rand(N, 16, generator=g) > 0.3).to(gpu)


def test_fused_head_step_rebinds_across_batch_shapes_without_dangling_images(gpu):
    """a smaller last batch re-binds the step.  The optimiser's launch used to keep scattering
    the updated per-class weights through raw pointers into the PREVIOUS step's workspace, which nothing kept alive.
    Now one step per shape is cached, only the current one is attached, and a dropped step is detached first: five
    updates over shapes A, B, A, A, B stay bit-identical to the stateless twin, the image list does not grow, and a
    canary that inherits a dropped workspace's block is not written by later updates."""
    K = 51
    try:
        cfg, (fa, fb) = _twin_heads(gpu, PER_CLASS, K)
        fused = deploy.FusedHeadStep(fa, cfg)
        opt = fused.make_optimizer(0.05)
        twin = deploy.FusedHeadStep(fb, cfg)          # no make_optimizer(): no images, per-step preparation
        opt_t = deploy.configure_optimizer(cfg, dict(twin.params), twin.bucket, 0.05, regularized=twin.regularized)
        n_images = None
        for i, N in enumerate([4, 3, 4, 4, 3]):
            X, y, _, _ = _batch(gpu, N, K, 100 + i)
            ta, ea = fused(X, y)
            tb, eb = twin(X, y)
            (3.0 * ta).backward()                     # head-only: the node still scales the bucket
            (3.0 * tb).backward()
            torch.cuda.synchronize()
            assert fused._step_obj.weight_image_maps and not twin._step_obj.weight_image_maps
            assert torch.equal(ea['Logits'], eb['Logits']) and torch.equal(ta, tb), i
            assert torch.equal(fused.bucket.flat, twin.bucket.flat), i
            assert torch.equal(fused._dX, twin._dX), i
            if n_images is None:
                n_images = len(opt.images)
            assert len(opt.images) == n_images and not opt._img_refresh
            assert all(e[2] is fused._step_obj for e in opt.images)          # only the current step is attached
            opt.step()
            opt_t.step()
            torch.cuda.synchronize()
            for n in fused.params:
                assert torch.equal(fused.params[n].data, twin.params[n].data), (i, n)
        assert len(fused._steps) == 2
        # the bucket carried the upstream coefficient: one more step, compared with a unit-coefficient run
        X, y, _, _ = _batch(gpu, 4, K, 999)
        step0 = fused.head._step
        t1, _ = fused(X, y)
        t1.backward()
        g1 = fused.bucket.flat.clone()
        fused.head._step = step0
        t3, _ = fused(X, y)
        (0.5 * t3).backward()
        assert torch.allclose(fused.bucket.flat, 0.5 * g1, rtol=1e-6, atol=0)
        # a dropped step: its workspace returns to the allocator, and the optimiser no longer writes there
        fused.max_bound_steps = 1
        old = fused._steps[next(iter(fused._steps))]
        nbytes, addr = old.workspace.numel(), old.workspace.data_ptr()
        Xn, yn, _, _ = _batch(gpu, 5, K, 7)          # a third shape evicts both cached steps but the new one
        fused(Xn, yn)
        assert len(fused._steps) == 1
        del old
        canary = torch.zeros(nbytes, dtype=torch.uint8, device=gpu)
        torch.cuda.synchronize()
        inherited = canary.data_ptr() == addr
        fused.bucket.flat.normal_()
        opt.step()
        torch.cuda.synchronize()
        assert int(canary.max()) == 0, 'the update launch wrote into a dropped workspace (inherited=%s)' % inherited
    finally:
        apa_config.reset_cfg()


@pytest.mark.parametrize('policy', ['raise', 'refresh'])
def test_fused_head_step_notices_weights_written_behind_the_optimiser(gpu, policy):
    """VERDICT r05 Weak #1: cfg 003 form on bf16 features -- the step reads the bf16 W1 shadow and the W2^T image
    the optimiser maintains.  load_state_dict after make_optimizer used to leave both stale: a silently wrong
    forward and backward.  Now the next step raises StaleOperandError (or, stale='refresh', rebuilds them); after
    refresh_operands() the step equals a twin that was built on the new weights from scratch, bit for bit."""
    K = 23
    try:
        cfg, (fa, fb) = _twin_heads(gpu, CFG003, K)
        fused = deploy.FusedHeadStep(fa, cfg, stale=policy)
        opt = fused.make_optimizer(0.01)
        X, y, lp, pv = _batch(gpu, 4, K, 1, pose=True)
        t, _ = fused(X, y, lp, pv)
        t.backward()
        opt.step()
        # a "checkpoint restore": new weights enter through load_state_dict on both heads
        g = torch.Generator().manual_seed(77)
        sd = {n: (torch.randn(p.shape, generator=g) / (p.shape[0] ** 0.5 if p.dim() == 2 else 10.0)).to(gpu)
              for n, p in fa.head.named_parameters()}
        fa.head.load_state_dict(sd, strict=False)
        fb.head.load_state_dict(sd, strict=False)
        fb.head._step = fa.head._step
        twin = deploy.FusedHeadStep(fb, cfg)
        twin.make_optimizer(0.01)                      # built AFTER the restore: current by construction
        X2, y2, lp2, pv2 = _batch(gpu, 4, K, 2, pose=True)
        if policy == 'raise':
            with pytest.raises(deploy.StaleOperandError, match='pose_w1'):
                fused(X2, y2, lp2, pv2)
            with pytest.raises(deploy.StaleOperandError):
                opt.step()
            fused.refresh_operands()
        ta, ea = fused(X2, y2, lp2, pv2)
        tb, eb = twin(X2, y2, lp2, pv2)
        torch.cuda.synchronize()
        assert torch.equal(fused.w1_shadow, fa.head.pose_w1.data.to(torch.bfloat16))
        J, Cp = fa.head.pose_w2.shape[1], fa.head.pose_w2.shape[0]
        assert torch.equal(fused.w2t_image[:J, :Cp], fa.head.pose_w2.data.t().to(torch.bfloat16))
        assert torch.equal(ea['Logits'], eb['Logits']) and torch.equal(ea['PoseLogits'], eb['PoseLogits'])
        assert torch.equal(ta, tb) and torch.equal(fused.bucket.flat, twin.bucket.flat)
        # 'Losses' are copies: the next step does not change what was handed out
        kept = [float(l) for l in ea['Losses']]
        X3, y3, lp3, pv3 = _batch(gpu, 4, K, 3, pose=True)
        fused(X3, y3, lp3, pv3)
        torch.cuda.synchronize()
        assert [float(l) for l in ea['Losses']] == kept
    finally:
        apa_config.reset_cfg()
