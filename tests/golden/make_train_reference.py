#!/usr/bin/env python
"""Golden vectors for the TRAINING STEP around the head, produced by
the REFERENCE'S OWN code: data-parallel clones, the clone-loss scaling and gradient sum, ITER_SIZE gradient
accumulation, the learning-rate schedule and the optimiser configuration.

What is executed from /root/reference (behind tests/golden/tf1_shim.py, float64):
  * models/slim/deployment/model_deploy.py  -- the whole module: DeploymentConfig, create_clones,
    _gather_clone_loss (clone loss / num_clones, regularisation losses on the FIRST clone only),
    _optimize_clone, optimize_clones, _sum_clones_gradients
  * src/train.py, taken out of the file by `ast` (the file cannot be imported: dataset / queue / session code):
      _configure_learning_rate (:29-69), _configure_optimizer (:72-105), _get_variables_to_train (:166-181),
      _train_step (:215-280: which of train_ops[i] a step runs),
      main()'s nested clone_fn (:393-424: dequeue -> network_fn -> gen_losses) and
      main()'s gradient-update block (:520-566: `if cfg.TRAIN.ITER_SIZE == 1: ... else: AccumulateGradients`)
  * models/slim/nets/nets_factory.py, src/loss.py, src/config.py  -- as for the head fixtures
What this script supplies itself (and restates from main(), train.py:296-518, because that code is one long
function around a real dataset and session): the DeploymentConfig arguments (:310-315), the global step, the
get_network_fn call (:343-349), `create_clones(deploy_config, clone_fn, [batch_queue])` (:429),
update_ops of the first clone (:433), the optimiser configuration call (:483-485) and
`optimize_clones(clones, optimizer, var_list=_get_variables_to_train(), clip_gradients=...)` (:507-511).
The batch queue hands out seeded synthetic batches (conv5 maps at the drop-in boundary instead of images).

How session.run is modelled is described in tf1_shim.py ("training graph").  One assumption about the TF1
runtime is made there and stated here: inside `with tf.control_dependencies([accumulate_op])` the read of the
`ref_grad` variables sees the accumulated value (train.py:551-553 relies on it).

tf.train.MomentumOptimizer / exponential_decay arithmetic is TensorFlow's (third-party): restated in the shim.

Run in the build container:   python tests/golden/make_train_reference.py
Output: tests/golden/ref_train_<case>.npz.  Test infrastructure only.
"""
from __future__ import annotations

import ast
import copy
import json
import os
import sys
import time
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf1_shim as tfs                                   # noqa: E402
import make_head_reference as mhr                        # noqa: E402

REF = mhr.REF
TRAIN_PY = os.path.join(REF, 'src', 'train.py')
f32 = mhr.f32
_rs = mhr._rs

P = mhr.P
TRAIN_CASES = [
    # cfg 002: two clones (GPUS '0,1'), the YAML's ITER_SIZE, dropout off so that the loop is deterministic
    dict(name='cfg002_2clones_iter2', yaml='002_MPII_ResNet_withAttention.yaml', gpus='0,1', steps=3, batch=2,
         shape=(1, 4, 4, 32), K=12, net={'DROPOUT': 0.0},
         train_cfg={'ITER_SIZE': 2, 'LEARNING_RATE': 0.05, 'NUM_STEPS_PER_DECAY': 2, 'WEIGHT_DECAY': 0.01}),
    # cfg 003 (pose loss on, pose-prelogit attention): one clone, three micro-steps per update
    dict(name='cfg003_1clone_iter3', yaml='003_MPII_ResNet_withPoseAttention.yaml', gpus='0', steps=2, batch=2,
         shape=(1, 3, 4, 16), K=10, net={'DROPOUT': 0.0},
         train_cfg={'ITER_SIZE': 3, 'LEARNING_RATE': 0.02, 'NUM_STEPS_PER_DECAY': 1, 'WEIGHT_DECAY': 0.005}),
    # ITER_SIZE 1: train_ops is the train tensor itself (train.py:520-528); decay from the epoch setting (:47-48)
    dict(name='cfg002_1clone_iter1_epoch_decay', yaml='002_MPII_ResNet_withAttention.yaml', gpus='0', steps=4,
         batch=3, shape=(1, 3, 3, 16), K=7, net={'DROPOUT': 0.0}, num_samples=12,
         train_cfg={'ITER_SIZE': 1, 'LEARNING_RATE': 0.1, 'NUM_STEPS_PER_DECAY': 0, 'NUM_EPOCHS_PER_DECAY': 0.5,
                    'BATCH_SIZE': 3, 'WEIGHT_DECAY': 0.02}),
    # dropout ON (the YAML's keep 0.2 rule is overridden to 0.5 to keep the tiny batch informative): every run
    # of every clone draws its own mask, recorded in order
    dict(name='cfg002_2clones_iter2_dropout', yaml='002_MPII_ResNet_withAttention.yaml', gpus='0,1', steps=2, batch=2,
         shape=(1, 4, 4, 32), K=12, net={'DROPOUT': 0.5},
         train_cfg={'ITER_SIZE': 2, 'LEARNING_RATE': 0.05, 'NUM_STEPS_PER_DECAY': 1, 'WEIGHT_DECAY': 0.01}),
]


# ------------------------------------------------------------------------------------ pieces of train.py
def _train_py_pieces():
    tree = ast.parse(open(TRAIN_PY).read(), TRAIN_PY)
    top = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    main = top['main']
    clone_fn = [n for n in ast.walk(main) if isinstance(n, ast.FunctionDef) and n.name == 'clone_fn']
    blocks = [n for n in ast.walk(main) if isinstance(n, ast.If) and ast.unparse(n.test) == 'cfg.TRAIN.ITER_SIZE == 1']
    assert len(clone_fn) == 1 and len(blocks) == 1
    return top, clone_fn[0], blocks[0]


def _exec_nodes(nodes, namespace):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, TRAIN_PY, 'exec'), namespace)
    return namespace


def load_training_reference():
    cfgmod, nf, lossmod = mhr.load_reference()
    tf = sys.modules['tensorflow']
    cfo = types.ModuleType('tensorflow.python.ops.control_flow_ops')
    cfo.with_dependencies = tfs.with_dependencies
    sys.modules['tensorflow.python.ops.control_flow_ops'] = cfo
    sys.modules['tensorflow.python.ops'].control_flow_ops = cfo
    md = mhr._exec_ref(os.path.join(REF, 'models', 'slim', 'deployment', 'model_deploy.py'), 'deployment.model_deploy')
    top, clone_fn_node, update_block = _train_py_pieces()
    ns = {'tf': tf, 'slim': tf.contrib.slim, 'cfg': cfgmod.cfg, 'np': np, 'time': time, 'os': os,
          'control_flow_ops': cfo, 'gen_losses': lossmod.gen_losses, '__name__': 'reftrain'}
    _exec_nodes([top['_configure_learning_rate'], top['_configure_optimizer'], top['_get_variables_to_train'],
                 top['_train_step']], ns)
    return cfgmod, nf, lossmod, md, ns, clone_fn_node, update_block


# ------------------------------------------------------------------------------------------------ a case
class BatchQueue(object):
    """slim.prefetch_queue stand-in: every dequeue() hands out the next seeded batch (and records it)."""

    def __init__(self, case, K, J, pose_hw):
        self.case, self.K, self.J, self.pose_hw = case, K, J, pose_hw
        self.batches = []
        self.probe = False

    def _make(self, i):
        c = self.case
        r = _rs(c['name'], 'batch', i)
        B = c['batch']
        T = c['shape'][0]
        images = f32(np.maximum(r.randn(B, *c['shape']), 0))                     # conv5 is post-ReLU
        labels_pose = f32(r.rand(B, T, self.pose_hw[0], self.pose_hw[1], self.J))
        valid = r.rand(B, T, self.J) > 0.3
        action = r.randint(0, self.K, size=(B,))
        return images, labels_pose, valid, action

    def dequeue(self):
        i = len(self.batches)
        b = self._make(i)
        if not self.probe:
            self.batches.append(b)
        images, labels_pose, valid, action = b
        return (tfs.Tensor(torch.from_numpy(images)), tfs.Tensor(torch.from_numpy(labels_pose)),
                tfs.Tensor(torch.from_numpy(valid)), tfs.Tensor(torch.from_numpy(action)))


def run_train_case(cfgmod, nf, lossmod, md, ns, clone_fn_node, update_block, defaults, case):
    name = case['name']
    mhr.reset_cfg(cfgmod, defaults)
    cfg = cfgmod.cfg
    cfgmod.cfg_from_file(os.path.join(REF, 'experiments', case['yaml']))
    mhr.merge(cfg.NET, case.get('net', {}))
    mhr.merge(cfg.TRAIN, case.get('train_cfg', {}))
    cfg.GPUS = case['gpus']
    K, J = case['K'], 16
    model = cfg.MODEL_NAME

    count = [0]

    def uniform_fn(shape, what):
        r = _rs(name, 'uniform', count[0])
        count[0] += 1
        return np.minimum(f32(r.random_sample(shape)), f32(1.0 - 2.0 ** -24))

    g = tfs.Graph(mhr.make_value_fn(name, 'trained'), uniform_fn)
    tfs.set_graph(g)

    def backbone_stub(imgs, num_classes, is_training=False, train_top_bn=False, **kw):
        return tfs.Tensor(torch.zeros(imgs.v.shape[0], num_classes, dtype=tfs.DT)), {nf.last_conv_map[model]: imgs}
    nf.networks_map[model] = backbone_stub

    # ---- train.py main(), :306-349
    num_clones = len(cfg.GPUS.split(','))
    deploy_config = md.DeploymentConfig(num_clones=num_clones, clone_on_cpu=False, replica_id=0, num_replicas=1,
                                        num_ps_tasks=0)
    global_step = tfs.GlobalStep()
    dataset = types.SimpleNamespace(num_classes=K, num_samples=case.get('num_samples', 1000))
    network_fn = nf.get_network_fn(cfg.MODEL_NAME, num_classes=dataset.num_classes, num_pose_keypoints=J,
                                   weight_decay=cfg.TRAIN.WEIGHT_DECAY, is_training=True, cfg=cfg)
    batch_queue = BatchQueue(case, K, J, case['shape'][1:3])

    # ---- the reference's clone_fn, with the names it closes over
    fn_ns = dict(ns, network_fn=network_fn, dataset=dataset)
    clone_fn = _exec_nodes([clone_fn_node], fn_ns)['clone_fn']

    # ---- :483-485
    learning_rate = ns['_configure_learning_rate'](dataset.num_samples, num_clones, global_step)
    optimizer = ns['_configure_optimizer'](learning_rate)

    current = {}

    def forward_backward():
        """:429 create_clones and :507-511 optimize_clones, evaluated on fresh batches"""
        g.begin_run()
        clones = md.create_clones(deploy_config, clone_fn, [batch_queue])
        first_clone_scope = deploy_config.clone_scope(0)
        update_ops = g.get_collection(tfs.GraphKeys.UPDATE_OPS, first_clone_scope)           # :433
        total_loss, gvs = md.optimize_clones(clones, optimizer, var_list=ns['_get_variables_to_train'](),
                                             clip_gradients=cfg.TRAIN.CLIP_GRADIENTS)
        current.update(total_loss=total_loss, gvs=gvs, update_ops=update_ops, clones=clones,
                       losses=[g.get_collection(tfs.GraphKeys.LOSSES, c.scope) for c in clones],
                       regs=g.get_collection(tfs.GraphKeys.REGULARIZATION_LOSSES))

    # graph construction: one evaluation for the structure (variables, which of them get a gradient); it does
    # not consume a batch or a random draw
    batch_queue.probe = True
    forward_backward()
    batch_queue.probe = False
    count[0] = 0
    del g.random_draws[:]
    assert not current['update_ops']                       # no batch-norm in these heads
    initial = {vn: g.variables[vn].detach().numpy().copy() for vn in g.var_order}
    grad_vars = [v for _g, v in current['gvs']]
    clones_gradients = [
        (tfs.Node((lambda i: lambda run: current['gvs'][i][0].v)(i), name=v.op.name + '/sum_grads',
                  shape=list(v.v.shape)), v) for i, v in enumerate(grad_vars)]
    total_loss = tfs.Node(lambda run: current['total_loss'].v, name='total_loss', shape=[])

    # ---- the reference's gradient-update block (:519-566)
    blk = dict(ns, optimizer=optimizer, clones_gradients=clones_gradients, global_step=global_step, update_ops=[],
               total_loss=total_loss, deploy_config=deploy_config, train_ops={})
    _exec_nodes([update_block], blk)
    train_ops = blk['train_ops']

    per_run = []

    def before_run():
        n0, d0 = len(batch_queue.batches), len(g.random_draws)
        forward_backward()
        assert [v.op.name for _g, v in current['gvs']] == [v.op.name for v in grad_vars]
        per_run.append(dict(batches=list(range(n0, len(batch_queue.batches))),
                            draws=list(range(d0, len(g.random_draws))),
                            grads={v.op.name: gr.v.detach().numpy().copy() for gr, v in current['gvs']},
                            values={v.op.name: v.v.detach().numpy().copy() for _gr, v in current['gvs']},
                            clone_losses=[[float(l.v.detach()) for l in ls] for ls in current['losses']],
                            reg_losses=[float(l.v.detach()) for l in current['regs']],
                            total_loss=float(current['total_loss'].v.detach())))
    sess = tfs.Session(before_run)
    lr_probe = tfs.Session(lambda: None)

    out, f32_keys = {}, []
    wd = float(cfg.TRAIN.WEIGHT_DECAY)

    def put(key, arr):
        arr = np.asarray(arr)
        if arr.dtype == np.float64 and arr.size > mhr.BIG:     # big tensors at float32 storage precision
            f32_keys.append(key)
            arr = arr.astype(np.float32)
        out[key] = arr

    steps = []
    for s in range(case['steps']):
        lr = float(lr_probe.run(learning_rate))
        r0 = len(per_run)
        total, should_stop = ns['_train_step'](sess, train_ops, global_step, {})                # the reference's step
        assert should_stop is False
        steps.append(dict(lr=lr, runs=list(range(r0, len(per_run))), total_loss=float(total),
                          global_step=int(global_step.value)))
        for vn in g.var_order:
            val = g.variables[vn].detach().numpy().copy()
            if val.size <= mhr.BIG or s in (0, case['steps'] - 1):          # big tensors: first and last update only
                put('step/%d/var/%s' % (s, vn), val)
    for vn, acc in optimizer.slots.items():
        put('final/momentum/' + vn, acc.numpy().copy())

    for vn, val in initial.items():
        out['var0/' + vn] = val.astype(np.float32)           # float32-representable by construction (make_value_fn)
        assert np.array_equal(out['var0/' + vn].astype(np.float64), val)
    for i, (images, labels_pose, valid, action) in enumerate(batch_queue.batches):
        out['batch/%d/images' % i] = images.astype(np.float32)
        out['batch/%d/labels_pose' % i] = labels_pose.astype(np.float32)
        out['batch/%d/labels_pose_valid' % i] = valid
        out['batch/%d/labels_action' % i] = action.astype(np.int64)
    draws = []
    for i, d in enumerate(g.random_draws):
        assert d['kind'] == 'dropout'
        keep = np.floor(d['keep_prob'] + d['uniform']).astype(np.uint8)
        out['rand/%d/keep_bits' % i] = np.packbits(keep.reshape(-1))
        draws.append({'kind': 'dropout', 'keep_prob': d['keep_prob'], 'shape': list(keep.shape)})
    # a variable that only the regulariser sees (the pose head of cfg 002): its summed clone gradient is
    # weight_decay * value in every run -- checked here, not stored
    reg_only = []
    for vn in [v.op.name for v in grad_vars]:
        if vn.endswith('/weights') and wd > 0 and all(
                np.array_equal(pr['grads'][vn], wd * pr['values'][vn]) for pr in per_run):
            reg_only.append(vn)
    for r, pr in enumerate(per_run):
        pr.pop('values')
        for vn, gr in pr.pop('grads').items():
            if vn not in reg_only:
                put('run/%d/grad/%s' % (r, vn), gr)
    net_flags = {k: v for k, v in cfg.NET.items() if not isinstance(v, dict)}
    meta = dict(case=name, model=model, num_classes=K, num_pose_keypoints=J, num_clones=num_clones,
                iter_size=int(cfg.TRAIN.ITER_SIZE), weight_decay=float(cfg.TRAIN.WEIGHT_DECAY), net=net_flags,
                train_cfg={k: cfg.TRAIN[k] for k in (
                    'LOSS_FN_POSE', 'LOSS_FN_POSE_WT', 'LOSS_FN_POSE_SAMPLED', 'LOSS_FN_ACTION', 'LOSS_FN_ACTION_WT',
                    'WEIGHT_DECAY', 'ITER_SIZE', 'LEARNING_RATE', 'LEARNING_RATE_DECAY_TYPE',
                    'LEARNING_RATE_DECAY_RATE', 'NUM_STEPS_PER_DECAY', 'NUM_EPOCHS_PER_DECAY', 'BATCH_SIZE',
                    'OPTIMIZER', 'MOMENTUM', 'CLIP_GRADIENTS', 'TRAINABLE_SCOPES')},
                num_samples=dataset.num_samples, optimizer=type(optimizer).__name__, momentum=optimizer.momentum,
                optimizer_applied=optimizer.applied, session_runs=sess.runs, var_order=g.var_order,
                grad_vars=[v.op.name for v in grad_vars], reg_only_grad=reg_only, f32_keys=f32_keys, var_init=g.var_init, steps=steps, runs=per_run,
                draws=draws, clone_scopes=[deploy_config.clone_scope(i) for i in range(num_clones)],
                train_ops=sorted(train_ops) if isinstance(train_ops, dict) else 'train_tensor')
    out['meta'] = np.array(json.dumps(meta, sort_keys=True, default=str))
    tfs.set_graph(None)
    return out


LR_GRID = [
    # TRAIN overrides, num_samples, num_clones
    ({'LEARNING_RATE_DECAY_TYPE': 'exponential', 'LEARNING_RATE': 0.01, 'NUM_STEPS_PER_DECAY': 0,
      'NUM_EPOCHS_PER_DECAY': 40.0, 'BATCH_SIZE': 16, 'ITER_SIZE': 2, 'LEARNING_RATE_DECAY_RATE': 0.33}, 15205, 4),
    ({'LEARNING_RATE_DECAY_TYPE': 'exponential', 'LEARNING_RATE': 0.001, 'NUM_STEPS_PER_DECAY': 300,
      'NUM_EPOCHS_PER_DECAY': 40.0, 'BATCH_SIZE': 10, 'ITER_SIZE': 1, 'LEARNING_RATE_DECAY_RATE': 0.1}, 1000, 1),
    ({'LEARNING_RATE_DECAY_TYPE': 'exponential', 'LEARNING_RATE': 0.02, 'NUM_STEPS_PER_DECAY': 0,
      'NUM_EPOCHS_PER_DECAY': 1.7, 'BATCH_SIZE': 7, 'ITER_SIZE': 3, 'LEARNING_RATE_DECAY_RATE': 0.5}, 999, 2),
    ({'LEARNING_RATE_DECAY_TYPE': 'fixed', 'LEARNING_RATE': 0.003, 'NUM_STEPS_PER_DECAY': 10,
      'NUM_EPOCHS_PER_DECAY': 2.0, 'BATCH_SIZE': 8, 'ITER_SIZE': 1, 'LEARNING_RATE_DECAY_RATE': 0.5}, 500, 1),
    ({'LEARNING_RATE_DECAY_TYPE': 'polynomial', 'LEARNING_RATE': 0.01, 'END_LEARNING_RATE': 0.0001,
      'NUM_STEPS_PER_DECAY': 50, 'NUM_EPOCHS_PER_DECAY': 2.0, 'BATCH_SIZE': 8, 'ITER_SIZE': 1,
      'LEARNING_RATE_DECAY_RATE': 0.5}, 500, 1),
]
LR_STEPS = [0, 1, 49, 50, 51, 299, 300, 1187, 1188, 5000]


def lr_schedule_table():
    """_configure_learning_rate (src/train.py:29-69) on LR_GRID x LR_STEPS."""
    cfgmod, _nf, _loss, _md, ns, _c, _u = load_training_reference()
    defaults = copy.deepcopy(cfgmod.cfg)
    rows = []
    sess = tfs.Session(lambda: None)
    for train_cfg, num_samples, num_clones in LR_GRID:
        mhr.reset_cfg(cfgmod, defaults)
        mhr.merge(cfgmod.cfg.TRAIN, train_cfg)
        gs = tfs.GlobalStep()
        lr = ns['_configure_learning_rate'](num_samples, num_clones, gs)
        for step in LR_STEPS:
            gs.value = torch.tensor(step, dtype=torch.int64)
            rows.append(dict(train_cfg=train_cfg, num_samples=num_samples, num_clones=num_clones, global_step=step,
                             lr=float(sess.run(lr))))
    mhr.reset_cfg(cfgmod, defaults)
    return rows


def generate(names=None):
    cfgmod, nf, lossmod, md, ns, clone_fn_node, update_block = load_training_reference()
    defaults = copy.deepcopy(cfgmod.cfg)
    res = {}
    for case in TRAIN_CASES:
        if names is None or case['name'] in names:
            res[case['name']] = run_train_case(cfgmod, nf, lossmod, md, ns, clone_fn_node, update_block, defaults, case)
    return res


if __name__ == '__main__':
    for name, blobs in generate().items():
        path = os.path.join(HERE, 'ref_train_%s.npz' % name)
        np.savez_compressed(path, **blobs)
        m = json.loads(str(blobs['meta']))
        print('%-36s %7d bytes  clones %d  iter %d  runs %d  applied %d  lr %s' % (
            name, os.path.getsize(path), m['num_clones'], m['iter_size'], m['session_runs'], m['optimizer_applied'],
            [round(s['lr'], 6) for s in m['steps']]))
    with open(os.path.join(HERE, 'ref_lr_schedule.json'), 'w') as f:
        json.dump(lr_schedule_table(), f, indent=0, sort_keys=True)
    print('wrote ref_lr_schedule.json')
