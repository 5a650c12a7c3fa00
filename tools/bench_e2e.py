#!/usr/bin/env python
"""End-to-end steps for BASELINE configs[1] / [2]: images -> torch-ROCm ResNet-101 (slim graph,
`attentionalpoolingaction_amd/resnet_v1.py`, channels-last) -> HIP attention head -> loss [-> backward -> update].

The reference runs the head as part of `network_fn(images)` (models/slim/nets/nets_factory.py:118-137,
src/eval.py:181, src/train.py:393-422); the head-only workloads of bench.py time the part this library
replaces, these two say what share of a real step that part is and where its input comes from:

  eval002   cfg 002 evaluation: fp32 backbone (eval-mode batch norm, no grad) -> attn-pool forward -> softmax +
            argmax (`eval_utils`), batch 32 x 448 x 448 x 3 -> conv5 32 x 14 x 14 x 2048.
  train003  cfg 003 training: backbone under bf16 autocast -> `deploy.FusedHeadStep` (the head's forward, pose L2 +
            softmax cross-entropy and backward as ONE host call, apa_pose_attn_train_step: 13 launches, conv5's
            gradient handed to autograd) -> backward through the backbone -> the head's fused momentum-SGD launch
            (deploy.MomentumSGD, which also rewrites the bf16 operand copy of W1; the L2 regulariser is its
            weight-decay term), torch.optim.SGD(momentum) on the backbone (src/train.py:90-94).
            `--module-path`: the same step through network_fn -> gen_losses -> autograd (the per-op calls) with
            torch.optim.SGD on the head -- what round 4 timed.

`build(...)` returns (step, info, probe): `step()` enqueues one step; `probe(n)` re-runs n steps with HIP events
around the head (forward: around the module call; backward: from just before `.backward()` to the hook on the
conv5 gradient) and returns the head's share of the step in device time.

Run directly (one JSON line), or under rocprofv3 (tools/profile_e2e.sh: kernel stats + FETCH_SIZE of the head's
forward streaming kernel inside the step, against the 51.4 MB it reads from HBM in the rotating-buffer bench)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG002 = {'MODEL_NAME': 'resnet_v1_101', 'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True,
                                                 'USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT': True},
          'TRAIN': {'LOSS_FN_POSE': '', 'LOSS_FN_ACTION': 'softmax-xentropy'}}
CFG003 = {'MODEL_NAME': 'resnet_v1_101', 'NET': {'USE_POSE_PRELOGITS_BASED_ATTENTION': True},
          'TRAIN': {'LOSS_FN_POSE': 'l2', 'LOSS_FN_ACTION': 'softmax-xentropy'}}


def build(which, dev, N=32, side=448, K=393, J=16, fuse_final_relu=False, backbone='resnet_v1_101',
          module_path=False):
    from attentionalpoolingaction_amd import config as apa_config, loss as apa_loss, nets_factory, deploy
    train = which == 'train003'
    cfg = apa_config.reset_cfg()
    apa_config.cfg_from_dict(dict(CFG003 if train else CFG002, MODEL_NAME=backbone))
    wd = float(cfg.TRAIN.WEIGHT_DECAY)
    torch.manual_seed(int(cfg.RNG_SEED))
    fn = nets_factory.get_network_fn(backbone, K, J, cfg, weight_decay=wd, is_training=train, device=dev,
                                     with_backbone=True, backbone_dtype=torch.bfloat16 if train else None,
                                     fuse_final_relu=fuse_final_relu)
    net, head = fn.backbone, fn.head
    g = torch.Generator().manual_seed(42)
    # vgg_preprocessing.py:355-372: RGB minus the channel means, no scaling -> roughly [-128, 128)
    images = (torch.rand(N, side, side, 3, generator=g) * 255.0 - 128.0).to(dev)
    labels = torch.randint(0, K, (N,), generator=g).to(dev)
    H = side // 32
    ev = {'head_fwd': None, 'head_bwd': None}

    # probe mode only (ev[...] set): events around the head module call, and on the gradient of its input
    def _pre(mod, args):
        if ev['head_fwd'] is not None:
            ev['head_fwd'][0].record()
            x = args[0]
            if ev['head_bwd'] is not None and x.requires_grad:
                x.register_hook(lambda g_, _e=ev['head_bwd']: _e[1].record())

    def _post(mod, args, out):
        if ev['head_fwd'] is not None:
            ev['head_fwd'][1].record()
    head.register_forward_pre_hook(_pre)
    head.register_forward_hook(_post)

    if not train:
        from attentionalpoolingaction_amd import eval_utils

        def step():
            with torch.no_grad():
                logits, _ = fn(images)
                return eval_utils.predict(logits)          # eval.py:193-197: softmax + argmax, one HIP launch
        params = None
    else:
        pose_lbl = torch.rand(N, H, H, J, generator=g).to(dev)
        valid = (torch.rand(N, J, generator=g) > 0.3).to(dev)
        bb_params = [p for p in net.parameters() if p.requires_grad]
        # slim's arg-scope regularises conv weights only (resnet_utils.py:241), never batch-norm beta / gamma
        decay = [p for p in bb_params if p.dim() == 4]
        nodecay = [p for p in bb_params if p.dim() != 4]
        lr, mom = float(cfg.TRAIN.LEARNING_RATE), float(cfg.TRAIN.MOMENTUM)
        opt_bb = torch.optim.SGD([{'params': decay, 'weight_decay': wd}, {'params': nodecay, 'weight_decay': 0.0}],
                                 lr=lr, momentum=mom, foreach=True)
        hp = [p for p in head.parameters() if p.requires_grad]
        reg = {id(w) for w in fn.regularized_weights()}
        if not module_path:
            fused = deploy.FusedHeadStep(fn, cfg, assume_unit_upstream=True)
            opt_fused = fused.make_optimizer(lr)

            def step():
                opt_bb.zero_grad(set_to_none=True)
                fused.probe_events = ev['head_fwd']
                total, _ = fused(images, labels, pose_lbl, valid)
                total.backward()              # conv5's gradient -> backbone; the head's gradients are in the bucket
                opt_fused.step()
                opt_bb.step()
                return total
        opt_head = torch.optim.SGD([{'params': [p for p in hp if id(p) in reg], 'weight_decay': wd},
                                    {'params': [p for p in hp if id(p) not in reg], 'weight_decay': 0.0}],
                                   lr=lr, momentum=mom, foreach=True)

        def step_module():
            logits, ep = fn(images)
            losses = apa_loss.gen_losses(labels, logits, cfg.TRAIN.LOSS_FN_ACTION, K, cfg.TRAIN.LOSS_FN_ACTION_WT,
                                         pose_lbl, ep['PoseLogits'], cfg.TRAIN.LOSS_FN_POSE, valid,
                                         cfg.TRAIN.LOSS_FN_POSE_WT, ep, cfg)
            # the L2 regulariser enters through the optimisers' weight_decay (same gradient, wd * w)
            total = sum(losses)
            opt_bb.zero_grad(set_to_none=True)
            opt_head.zero_grad(set_to_none=True)
            if ev['head_bwd'] is not None:
                ev['head_bwd'][0].record()
            total.backward()
            opt_head.step()
            opt_bb.step()
            return total
        if module_path:
            step = step_module
        params = sum(p.numel() for p in bb_params)

    def probe(n=5):
        """device time of the head inside the step: (head_fwd_ms, head_bwd_ms, step_ms), averages over n steps"""
        pairs = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        tf_, tb_, ts_ = [], [], []
        for _ in range(n):
            ev['head_fwd'], ev['head_bwd'] = pairs(), (pairs() if (train and module_path) else None)
            s0, s1 = pairs()
            s0.record()
            step()
            s1.record()
            torch.cuda.synchronize()
            tf_.append(ev['head_fwd'][0].elapsed_time(ev['head_fwd'][1]))
            if train and module_path:
                tb_.append(ev['head_bwd'][0].elapsed_time(ev['head_bwd'][1]))
            ts_.append(s0.elapsed_time(s1))
        ev['head_fwd'] = ev['head_bwd'] = None
        avg = lambda v: sum(v) / len(v) if v else 0.0
        return avg(tf_), avg(tb_), avg(ts_)

    info = {'workload': (('cfg003 training step end to end: {bb} (bf16 autocast, channels-last, torch-ROCm/MIOpen) -> HIP '
                          'pose head + attention head (bf16) -> pose L2 + softmax-xent -> backward through head and '
                          'backbone -> momentum-SGD; batch {n} x {s}x{s}x3 -> conv5 {n} x {h}x{h}x2048, K={k}' +
                          ('; head through the per-op module path' if module_path else
                           '; head forward + losses + backward as ONE host call (deploy.FusedHeadStep), fused HIP '
                           'momentum-SGD on the head'))
                         if train else
                         'cfg002 evaluation step end to end: {bb} (fp32, channels-last, torch-ROCm/MIOpen, no grad) -> HIP '
                         'attn-pool forward -> argmax; batch {n} x {s}x{s}x3 -> conv5 {n} x {h}x{h}x2048, K={k}'
                         ).format(bb=backbone, n=N, s=side, h=H, k=K) + ('; last ReLU folded into the op' if fuse_final_relu else ''),
            'N': N, 'dtype': 'bf16' if train else 'f32', 'backbone_params': params,
            'fused_head': bool(train and not module_path)}
    return step, info, probe


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    per = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / steps)
    per.sort()
    return per[1]


def run(which, dev, steps=6, warmup=3, **kw):
    step, info, probe = build(which, dev, **kw)
    sec = timed(step, steps, warmup)
    hf, hb, st = probe(4)
    N = info['N']
    out = {'workload': info['workload'], 'images_per_sec': round(N / sec, 1), 'ms_per_step': round(sec * 1e3, 3),
           'dtype': info['dtype'],
           'head_share': {'head_fwd_ms': round(hf, 4), 'head_bwd_ms': round(hb, 4), 'step_ms_under_probe': round(st, 3),
                          'fraction_of_step': round((hf + hb) / st, 5) if st > 0 else None,
                          'how': ('HIP events on the compute stream around the ONE call that is the head\'s forward, '
                                  'losses and backward (reported as head_fwd_ms; head_bwd_ms = 0)') if info['fused_head']
                          else ('HIP events on the compute stream: around the head module call (forward), and from '
                                'just before .backward() to the autograd hook on the conv5 gradient (backward: '
                                'loss kernels + head backward)')}}
    from attentionalpoolingaction_amd import config as apa_config
    apa_config.reset_cfg()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='eval002', choices=['eval002', 'train003'])
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--side', type=int, default=448)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--fuse-final-relu', action='store_true')
    ap.add_argument('--backbone', default='resnet_v1_101')
    ap.add_argument('--module-path', action='store_true',
                    help='train003: drive the head through network_fn -> gen_losses -> autograd (per-op calls)')
    args = ap.parse_args()
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    cof.load_library()
    dev = torch.device('cuda:0')
    print(json.dumps(run(args.workload, dev, args.steps, args.warmup, N=args.batch, side=args.side,
                         fuse_final_relu=args.fuse_final_relu, backbone=args.backbone,
                         module_path=args.module_path)))


if __name__ == '__main__':
    main()
