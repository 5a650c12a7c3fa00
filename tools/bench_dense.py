#!/usr/bin/env python
"""Supplementary measurements for the MFMA-bound rows of SURVEY.md section 8 (NOT the headline
bench -- that is /bench.py on the cfg 002 workload):

  cfg003    pose-regularised attention, bf16 (BASELINE configs[2]): PoseLogits head fwd (X.W1+relu,
            Ppre.W2) -> attention from pose_pre_logits -> pose L2 + softmax-xent -> backward of
            both heads.  Dense work: 1.864 GFLOP/img in the pose head.
  perclass  per-class bottom-up maps (M == K), HMDB-51 shape (K = 51, bf16) or K = 393:
            1.893 GFLOP/img at K = 393, 0.246 at K = 51.

Prints one JSON line per workload with images/sec and the achieved TFLOP/s of the whole step
against the dense bf16 MFMA peak (2.5 PFLOP/s) / the fp32 MFMA peak (157.3 TFLOP/s).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof  # noqa: E402

PEAK = {'bf16': 2500.0, 'f32': 157.3}  # TFLOP/s, dense (MI355X_MICROARCH.md)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    import sys
    print('host enqueue {:.1f} us/step, wall {:.1f} us/step'.format((t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6),
          file=sys.stderr)
    return (t2 - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cfg003', choices=['cfg003', 'perclass', 'eval002'])
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--hw', type=int, default=14)
    ap.add_argument('--classes', type=int, default=None)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-rank1', action='store_true',
                    help='cfg003: materialise the [N,P,768] attention-branch gradient between the two backward calls')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    N, H, C, Cp, J = args.batch, args.hw, 2048, 768, 16
    P = H * H
    td = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(42)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(td).to(dev)

    if args.workload == 'cfg003':
        K = args.classes or 393
        W1 = (torch.randn(C, Cp, generator=g) / C ** 0.5).to(dev); b1 = torch.zeros(Cp, device=dev)
        W2 = (torch.randn(Cp, J, generator=g) / Cp ** 0.5).to(dev); b2 = torch.zeros(J, device=dev)
        Wa = (torch.randn(Cp, 1, generator=g) / Cp ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
        Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
        labels = torch.randint(0, K, (N,), generator=g).to(dev)
        lbl = torch.rand(N, P, J, generator=g).to(dev)
        valid = (torch.rand(N, J, generator=g) > 0.3).to(dev)
        flags = cof.attn_flags(False, False, True)
        ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        pws = aws = None
        wa_flat = Wa.view(-1)

        def step():
            nonlocal pws, aws
            Ppre, Pl, pws = cof.pose_head_fwd(X, W1, b1, W2, b2, workspace=pws)
            logits, att, zs, ab, _, aws = cof.attn_pool_fwd(X, Ppre, Wa, ba, Wt, bt, flags=flags, keep_prob=0.2,
                                                           seed=42, offset=ctr, workspace=aws)
            _, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels)
            _, dPl = cof.pose_l2_loss_fwd_bwd(Pl, lbl, valid)
            # the attention-branch gradient crosses to the pose head in rank-1 form (dZ, wa)
            dX, dZ, *_ = cof.attn_pool_bwd(X, Ppre, Wa, ba, Wt, bt, att, zs, ab, G, flags=flags, keep_prob=0.2,
                                           seed=42, offset=ctr, workspace=aws, dxatt_rank1=not args.no_rank1)
            if args.no_rank1:
                cof.pose_head_bwd(X, W1, W2, Ppre, dPl, dZ, dX=dX, accumulate_dX=True, workspace=pws)
            else:
                cof.pose_head_bwd(X, W1, W2, Ppre, dPl, None, dX=dX, accumulate_dX=True, workspace=pws,
                                  ext_rank1=(dZ, wa_flat))

        flops_img = 3 * (2.0 * P * C * Cp + 2.0 * P * Cp * J)          # fwd + 2x bwd
        name = 'cfg003 pose-regularised attention head fwd+bwd (pose head 2048->768->16 + M=1 pooling)'
    elif args.workload == 'eval002':
        # BASELINE configs[1]: cfg 002 evaluation, forward + softmax probabilities + argmax (eval.py:181-197)
        # as one host call; HBM-bound (reads X once): reported against the 8 TB/s peak
        K = args.classes or 393
        Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
        Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
        ev = cof.HeadEvalStep(X, X, Wa, ba, Wt, bt)
        step = ev.run
        sec = timed(step, args.steps, args.warmup)
        gbs = N * P * C * X.element_size() / sec / 1e9
        print(json.dumps({
            'workload': 'cfg002 eval step (attn-pool forward + softmax + argmax, one call); per-GPU batch '
                        '{} x {}x{}x{} {}, K={}'.format(N, H, H, C, args.dtype, K),
            'images_per_sec': round(N / sec, 1), 'ms_per_step': round(sec * 1e3, 4),
            'roofline': {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': round(gbs / 8000.0, 4), 'note': 'whole step: P*C*s bytes per image'}}))
        return
    else:
        K = args.classes or 51
        Wa = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(K, device=dev)
        Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
        labels = torch.randint(0, K, (N,), generator=g).to(dev)
        flags = cof.attn_flags(False, False, True)
        ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        grads = (torch.empty_like(X), None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt),
                 torch.empty_like(bt))
        st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, flags=flags, keep_prob=0.2, seed=42,
                               offset=ctr)                       # one host call per step
        step = st.run

        flops_img = 3 * (2 * 2.0 * P * C * K)                           # Z and T products, fwd + 2x bwd
        name = 'per-class bottom-up maps (M=K) attention head fwd+bwd'

    sec = timed(step, args.steps, args.warmup)
    tflops = N * flops_img / sec / 1e12
    print(json.dumps({
        'workload': '{}; per-GPU batch {} x {}x{}x{} {}, K={}'.format(name, N, H, H, C, args.dtype, K),
        'images_per_sec': round(N / sec, 1), 'ms_per_step': round(sec * 1e3, 4),
        'roofline': {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': PEAK[args.dtype],
                     'unit': 'TFLOP/s', 'frac': round(tflops / PEAK[args.dtype], 4),
                     'algorithmic_gflop_per_image': round(flops_img / 1e9, 3),
                     'note': 'whole step (all kernels) against the dense MFMA peak'}}))


if __name__ == '__main__':
    main()
