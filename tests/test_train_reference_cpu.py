"""The TRAINING STEP around the head against vectors produced by the
reference's own model_deploy.py and the training pieces of src/train.py (tests/golden/make_train_reference.py
lists exactly what is executed and what is restated):

  * the product's HOST logic -- deploy.DeploymentConfig.clone_loss_scale, the gradient sum over clones,
    deploy.GradientAccumulator (ITER_SIZE), deploy.configure_learning_rate / decay_steps,
    deploy.configure_optimizer -> MomentumSGD with the L2 term folded in -- driven with the CPU oracle's float64
    gradients, lands on the reference's variables after every parameter update
  * per session.run the oracle's summed clone gradient == the reference's `clones_gradients`
  * graph facts: which run applies, how many batches a run dequeues, where the regulariser is counted
The same replay with the HIP head and the fused HIP optimiser is tests/test_train_reference_gpu.py.
"""
import json
import os

import numpy as np
import pytest
import torch

import _ref_fixture as rf
from attentionalpoolingaction_amd import config as apa_config, deploy

TRAIN_PATHS = rf.train_fixture_paths()


def _id(path):
    return os.path.basename(path)[len('ref_train_'):-4]


def product_cfg(tf_):
    apa_config.reset_cfg()
    net = {k: v for k, v in tf_.meta['net'].items() if k != 'USE_POSE_ATTENTION_LOGITS_DIMS'}
    return apa_config.cfg_from_dict({'MODEL_NAME': tf_.meta['model'], 'NET': net, 'TRAIN': dict(tf_.meta['train_cfg'])})


def oracle_clone_gradients(tf_, variables, batch, draw):
    """-> ({tf name: d(sum of the clone's LOSSES)/d(variable)}, [losses]) in float64, unscaled, no regulariser"""
    fx = tf_.clone_fixture(variables, batch, draw, weight_decay=0.0)
    o = rf.run_oracle(fx)
    return {vn: o['grad/var/' + vn] for vn in tf_.meta['var_order']}, o['out/losses']


def replay(tf_, clone_gradients, dtype=torch.float64, device='cpu', on_run=None):
    """The training loop of src/train.py + model_deploy.py written with the PRODUCT's deploy pieces.
    `clone_gradients(params, batch, draw) -> ({name: grad}, losses)` supplies one clone's gradient."""
    m = tf_.meta
    cfg = product_cfg(tf_)
    names = m['var_order']
    params = {vn: torch.from_numpy(v).to(dtype).to(device) for vn, v in tf_.initial_variables().items()}
    bucket = deploy.GradientBucket({vn: params[vn].shape for vn in names}, device, dtype=dtype)
    accum = deploy.GradientAccumulator(bucket, cfg.TRAIN.ITER_SIZE)
    regularized = [vn for vn in names if vn.endswith('/weights')]            # slim.l2_regularizer: conv weights
    opt = deploy.configure_optimizer(cfg, params, bucket, cfg.TRAIN.LEARNING_RATE, regularized=regularized)
    global_step = 0
    history = []
    for s, step in enumerate(m['steps']):
        lr = deploy.configure_learning_rate(cfg, m['num_samples'], m['num_clones'], global_step)
        assert abs(lr - step['lr']) <= 1e-12 * step['lr']
        for r in step['runs']:
            run = m['runs'][r]
            assert len(run['batches']) == m['num_clones']                   # one dequeue per clone per run
            bucket.zero_()
            clone_losses = []
            for ci, (b, d) in enumerate(zip(run['batches'], run['draws'])):
                dc = deploy.DeploymentConfig(num_clones=m['num_clones'], clone_index=ci)
                grads, losses = clone_gradients(params, b, d)
                clone_losses.append(losses)
                for vn in names:                                            # the all-reduce(SUM) of the buckets
                    bucket.views[vn].add_(torch.as_tensor(grads[vn]).to(dtype).to(device).reshape(params[vn].shape),
                                          alpha=dc.clone_loss_scale)
            if on_run is not None:
                on_run(r, run, params, bucket, clone_losses)
            applied = accum.step()
            assert applied == (r == step['runs'][-1])                        # train.py:558-566 / _train_step
            if applied:
                opt.step(lr=lr)
                global_step += 1
        assert global_step == step['global_step']
        history.append({vn: params[vn].detach().cpu().double().numpy().copy() for vn in names})
    apa_config.reset_cfg()
    return history, opt


def test_fixture_inventory():
    names = [_id(p) for p in TRAIN_PATHS]
    assert len(names) >= 4
    metas = [rf.TrainFixture(p).meta for p in TRAIN_PATHS]
    assert {m['num_clones'] for m in metas} == {1, 2} and {m['iter_size'] for m in metas} >= {1, 2, 3}
    assert any(float(m['net']['DROPOUT']) > 0 for m in metas)


@pytest.mark.parametrize('path', TRAIN_PATHS, ids=_id)
def test_reference_training_graph_facts(path):
    tf_ = rf.TrainFixture(path)
    m = tf_.meta
    it, nc = m['iter_size'], m['num_clones']
    assert m['optimizer'] == 'MomentumOptimizer' and m['momentum'] == 0.9          # cfgs 002/003, train.py:90-94
    assert m['session_runs'] == it * len(m['steps'])                               # _train_step: ITER_SIZE runs / step
    assert m['optimizer_applied'] == len(m['steps'])                               # ... of which the last applies
    assert m['train_ops'] == ('train_tensor' if it == 1 else list(range(it)))
    assert m['clone_scopes'] == (['clone_%d' % i for i in range(nc)] if nc > 1 else [''])
    for step in m['steps']:
        assert len(step['runs']) == it
        last = m['runs'][step['runs'][-1]]
        assert abs(step['total_loss'] - last['total_loss']) < 1e-12                # the step reports the LAST run's loss
    for run in m['runs']:
        # total_loss = sum_clones( sum(clone LOSSES) / num_clones ) + regularisation losses ONCE (first clone)
        want = sum(sum(ls) for ls in run['clone_losses']) / nc + sum(run['reg_losses'])
        assert abs(run['total_loss'] - want) <= 1e-12 * abs(want)
    # variables that get no gradient on any clone (tf.gradients -> None) are not in clones_gradients at all
    # (_sum_clones_gradients drops them): for cfg 002 the pose biases
    untouched = [vn for vn in m['var_order'] if vn not in m['grad_vars']]
    for vn in untouched:
        assert vn.endswith('/biases')
        last = len(m['steps']) - 1
        assert np.array_equal(tf_.arrays['step/%d/var/%s' % (last, vn)], tf_.arrays['var0/' + vn].astype(np.float64))


@pytest.mark.parametrize('path', TRAIN_PATHS, ids=_id)
def test_product_deploy_logic_with_oracle_gradients_is_the_reference_loop(path):
    tf_ = rf.TrainFixture(path)
    m = tf_.meta
    wd = m['weight_decay']

    def clone_gradients(params, b, d):
        return oracle_clone_gradients(tf_, {vn: p.numpy() for vn, p in params.items()}, b, d)

    def on_run(r, run, params, bucket, clone_losses):
        # the oracle's summed clone gradient (+ the regulariser's wd * w, which the reference adds on clone 0)
        for vn in m['grad_vars']:
            got = bucket.views[vn].numpy() + (wd * params[vn].numpy() if vn.endswith('/weights') else 0.0)
            key = 'run/%d/grad/%s' % (r, vn)
            exp = wd * params[vn].numpy() if vn in m['reg_only_grad'] else tf_.arrays[key]
            scale = max(np.abs(exp).max(), 1e-30)
            assert np.abs(got - exp).max() <= tf_.tol(key) * scale + 1e-15, key
        for ls, exp in zip(clone_losses, run['clone_losses']):
            assert np.allclose(ls, exp, rtol=1e-12, atol=0)

    history, opt = replay(tf_, clone_gradients, on_run=on_run)
    for s, vars_ in enumerate(history):
        for vn, got in vars_.items():
            key = 'step/%d/var/%s' % (s, vn)
            if key not in tf_.arrays:
                continue
            exp = tf_.arrays[key]
            assert np.abs(got - exp).max() <= tf_.tol(key) * max(np.abs(exp).max(), 1e-30) + 1e-15, key
    # the momentum slots (the product folds wd * w into the slot, exactly like a regularisation-loss gradient)
    o = 0
    for vn in opt.bucket.names:
        n = opt.params[vn].numel()
        key = 'final/momentum/' + vn
        if key in tf_.arrays:
            exp = tf_.arrays[key]
            got = opt.acc[o:o + n].view_as(opt.params[vn]).numpy()
            assert np.abs(got - exp).max() <= tf_.tol(key) * max(np.abs(exp).max(), 1e-30) + 1e-15, key
        o += n


def test_learning_rate_table_matches_the_reference():
    """_configure_learning_rate (src/train.py:29-69) called by the generator for a grid of configurations and
    global steps: exponential / fixed / polynomial, step- and epoch-driven decay."""
    table = json.load(open(os.path.join(rf.GOLD, 'ref_lr_schedule.json')))
    assert len(table) >= 30
    kinds = set()
    for row in table:
        cfg = apa_config.reset_cfg()
        apa_config.cfg_from_dict({'TRAIN': row['train_cfg']})
        kinds.add(row['train_cfg']['LEARNING_RATE_DECAY_TYPE'])
        got = deploy.configure_learning_rate(cfg, row['num_samples'], row['num_clones'], row['global_step'])
        assert abs(got - row['lr']) <= 1e-12 * abs(row['lr']), row
    assert kinds == {'exponential', 'fixed', 'polynomial'}
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict({'TRAIN': {'LEARNING_RATE_DECAY_TYPE': 'cosine'}})
    with pytest.raises(ValueError):
        deploy.configure_learning_rate(cfg, 100, 1, 0)
    apa_config.reset_cfg()


@pytest.mark.regen
def test_generator_reproduces_a_committed_training_fixture():
    import importlib.util
    import sys
    saved, saved_path = dict(sys.modules), list(sys.path)
    try:
        spec = importlib.util.spec_from_file_location('make_train_reference',
                                                      os.path.join(rf.GOLD, 'make_train_reference.py'))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        out = gen.generate(names=['cfg002_2clones_iter2_dropout'])['cfg002_2clones_iter2_dropout']
        d = np.load(os.path.join(rf.GOLD, 'ref_train_cfg002_2clones_iter2_dropout.npz'))
        assert set(out) == set(d.files)
        for k in d.files:
            if k == 'meta':
                assert json.loads(str(out[k])) == json.loads(str(d[k]))
            else:
                assert np.array_equal(out[k], d[k]), k
        assert gen.lr_schedule_table() == json.load(open(os.path.join(rf.GOLD, 'ref_lr_schedule.json')))
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]


# ---------------------------------------------------------------------------------------------------------------
# the same reference loop with the clones as PROCESSES: world_size-2 gloo, one rank per clone of the reference
# ---------------------------------------------------------------------------------------------------------------
def _gloo_clone_worker(rank, world, port, path, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        tf_ = rf.TrainFixture(path)
        m = tf_.meta
        cfg = product_cfg(tf_)
        dc = deploy.DeploymentConfig()                                     # rank / world from the process group
        assert dc.num_clones == m['num_clones'] == world and dc.clone_index == rank
        names = m['var_order']
        params = {vn: torch.from_numpy(v) for vn, v in tf_.initial_variables().items()}
        bucket = deploy.GradientBucket({vn: params[vn].shape for vn in names}, 'cpu', dtype=torch.float64)
        accum = deploy.GradientAccumulator(bucket, cfg.TRAIN.ITER_SIZE)
        opt = deploy.configure_optimizer(cfg, params, bucket, cfg.TRAIN.LEARNING_RATE,
                                         regularized=[vn for vn in names if vn.endswith('/weights')])
        step_no = 0
        for step in m['steps']:
            lr = deploy.configure_learning_rate(cfg, m['num_samples'], world, step_no)
            for r in step['runs']:
                run = m['runs'][r]
                grads, _ = oracle_clone_gradients(tf_, {vn: p.numpy() for vn, p in params.items()},
                                                  run['batches'][rank], run['draws'][rank])   # THIS clone's batch
                for vn in names:
                    bucket.views[vn].copy_(torch.from_numpy(np.asarray(grads[vn])).reshape(params[vn].shape)
                                           * dc.clone_loss_scale)
                if accum.step():                    # ITER_SIZE local micro-steps, ONE all-reduce per update
                    deploy.sum_clone_gradients(bucket, dc)
                    opt.step(lr=lr)
                    step_no += 1
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **{vn: p.numpy() for vn, p in params.items()})
    finally:
        dist.destroy_process_group()
        apa_config.reset_cfg()


@pytest.mark.parametrize('name', ['cfg002_2clones_iter2', 'cfg002_2clones_iter2_dropout'])
def test_two_gloo_ranks_replay_the_reference_two_clone_loop(tmp_path, name):
    """Each rank plays one clone of the reference run (its own batches and dropout masks, oracle gradients),
    accumulates ITER_SIZE micro-steps LOCALLY and all-reduces once per update (the reference sums the
    clone gradients every run and accumulates the sums -- the same numbers, one collective instead of ITER_SIZE).
    Both ranks must hold the reference's variables after the last update."""
    import socket
    import torch.multiprocessing as mp
    path = os.path.join(rf.GOLD, 'ref_train_%s.npz' % name)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_clone_worker, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    tf_ = rf.TrainFixture(path)
    last = len(tf_.meta['steps']) - 1
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    for vn in tf_.meta['var_order']:
        assert np.array_equal(r0[vn], r1[vn]), vn                          # replicas stay identical
        key = 'step/%d/var/%s' % (last, vn)
        exp = tf_.arrays[key]
        assert np.abs(r0[vn] - exp).max() <= tf_.tol(key) * max(np.abs(exp).max(), 1e-30) + 1e-15, key
