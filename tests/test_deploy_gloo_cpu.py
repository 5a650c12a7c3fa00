"""N > 1 path on CPU: world_size-2 `gloo` process group exercising deploy.py (flat gradient bucket,
all-reduce of tower gradients, loss pre-scaling, regularisation added once, ITER_SIZE accumulation,
momentum SGD) against the oracle's restatement of model_deploy.py."""
import os
import socket

import numpy as np
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


def test_exponential_decay_staircase():
    # cfg 002: lr 1e-3, x0.33 every 5000 steps (experiments/002...yaml:9-14)
    f = lambda s: deploy.exponential_decay_lr(1e-3, s, 5000, 0.33)
    assert f(0) == f(4999) == 1e-3
    assert f(5000) == 1e-3 * 0.33 and abs(f(11999) - 1e-3 * 0.33 ** 2) < 1e-18
    assert deploy.exponential_decay_lr(1.0, 2500, 5000, 0.25, staircase=False) == 0.5
