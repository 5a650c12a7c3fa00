#!/usr/bin/env python
"""Supplementary workloads for the MFMA-bound rows of SURVEY.md section 8 and the evaluation step.
`/bench.py` imports the builders below and reports them in the `extra` object of its JSON line (the
headline stays the cfg 002 training workload); run this file directly for one workload at a time:

  cfg003    pose-regularised attention, bf16 (BASELINE configs[2]): PoseLogits head fwd (X.W1+relu,
            Ppre.W2) -> attention from pose_pre_logits -> pose L2 + softmax-xent -> backward of
            both heads.  Dense work: 1.864 GFLOP/img in the pose head.
  perclass  per-class bottom-up maps (M == K), HMDB-51 shape (K = 51, bf16) or K = 393:
            1.893 GFLOP/img at K = 393, 0.246 at K = 51.
  eval002   BASELINE configs[1]: cfg 002 evaluation step (forward + softmax + argmax, one call).

Each builder returns (step_fn, info): `step_fn()` enqueues one step on the current stream; `info`
carries the workload name and the algorithmic work per image.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK = {'bf16': 2500.0, 'f32': 157.3}  # TFLOP/s, dense (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


L3_BYTES = 256 << 20    # Infinity Cache (MI355X_MICROARCH.md)


def _features(N, P, C, dtype, dev, seed=42):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.relu(torch.randn(N, P, C, generator=g, device=dev)).to(dtype)


def _n_sets(per_set_bytes, rotate):
    """how many (X, dX) buffer sets a workload cycles through: enough that 1.5 x the 256 MiB Infinity Cache lies
    between two uses of the same set (bench.py's headline rule), so that no step finds its feature map cached from
    the previous one.  rotate = 1: one set (kernel-by-kernel profiling of a variant)."""
    if rotate and rotate > 0:
        return int(rotate)
    return max(3, -(-int(1.5 * L3_BYTES) // int(per_set_bytes)))


def _round_robin(runs):
    state = {'i': 0}
    n = len(runs)

    def step():
        runs[state['i'] % n]()
        state['i'] += 1
    return step


def _rot_note(R, per_set_bytes):
    return '; X/dX rotated over {} sets ({:.0f} MB live; workspace and activations shared, as in a training loop)'.format(R, R * per_set_bytes / 1e6) if R > 1 else \
        '; ONE buffer set (features stay in the Infinity Cache between steps)'


def build_cfg003(cof, dev, N=32, H=14, K=393, dtype='bf16', rank1=True, one_call=True, rotate=0):
    C, Cp, J, P = 2048, 768, 16, H * H
    td = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(42)
    per_set = 2 * N * P * C * (2 if dtype == 'bf16' else 4)
    R = _n_sets(per_set, rotate) if (one_call and rank1) else 1
    X = _features(N, P, C, td, dev)
    W1 = (torch.randn(C, Cp, generator=g) / C ** 0.5).to(dev); b1 = torch.zeros(Cp, device=dev)
    W2 = (torch.randn(Cp, J, generator=g) / Cp ** 0.5).to(dev); b2 = torch.zeros(J, device=dev)
    Wa = (torch.randn(Cp, 1, generator=g) / Cp ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
    labels = torch.randint(0, K, (N,), generator=g).to(dev)
    lbl = torch.rand(N, P, J, generator=g).to(dev)
    valid = (torch.rand(N, J, generator=g) > 0.3).to(dev)
    flags = cof.attn_flags(False, False, True)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    state = {'pws': None}
    wa_flat = Wa.view(-1)
    # caller-owned activations and gradients: the attention part of the step (pooling forward, softmax
    # cross-entropy, pooling backward) is ONE host call bound to them (cof.HeadTrainStep, the call the
    # headline configuration uses); its attention input is the pose head's Ppre, its attention-branch
    # gradient crosses back to the pose head in rank-1 form (dZ, wa)
    Ppre = torch.empty((N, P, Cp), dtype=td, device=dev)
    Pl = torch.empty((N, P, J), dtype=torch.float32, device=dev)
    dX = torch.empty_like(X)
    dZ = torch.empty((N * P,), dtype=torch.float32, device=dev)
    grads = (dX, dZ, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt), torch.empty_like(bt))
    if one_call and rank1:
        # the whole step as ONE host call (apa_pose_attn_train_step): neighbouring ops share launches, and the bf16
        # operand copy of W1 is the caller's -- in a training loop the optimizer's own launch keeps it current
        # (apa_momentum_sgd_step_shadow), here the weights do not change between steps
        w1_bf16 = W1.to(torch.bfloat16).contiguous()
        w2t_bf16 = cof.pose_w2t_image(W2)      # ... and so is the bf16 image of W2^T the Pl product reads
        new = lambda t: torch.empty_like(t)
        pgrads = (new(W1), new(b1), new(W2), new(b2), new(Wa), new(ba), new(Wt), new(bt))
        sts = []
        for r in range(R):        # one bound step per (X, dX) set; everything else is shared, as in bench.py's headline:
            #                       weights, parameter gradients, activations, workspaces, dropout counter
            Xr = X if r == 0 else _features(N, P, C, td, dev, seed=42 + r)
            sts.append(cof.PoseAttnTrainStep(Xr, (W1, b1, W2, b2, Wa, ba, Wt, bt), labels, lbl, valid,
                                             (dX if r == 0 else new(Xr),) + pgrads,
                                             flags=flags, keep_prob=0.2, seed=42, offset=ctr, w1_bf16=w1_bf16,
                                             w2t_bf16=w2t_bf16, share_with=sts[0] if sts else None))
        info = {'workload': 'cfg003 pose-regularised attention head fwd+bwd (pose head 2048->768->16 + M=1 '
                            'pooling + pose L2 + softmax-xent), one host call; per-GPU batch {} x {}x{}x{} {}, K={}, '
                            'dropout keep=0.2'.format(N, H, H, C, dtype, K) + _rot_note(R, per_set),
                'bound': 'mfma', 'dtype': dtype, 'N': N, 'rotate': R,
                'flops_per_image': 3 * (2.0 * P * C * Cp + 2.0 * P * Cp * J)}
        return _round_robin([st.run for st in sts]), info
    if rank1:
        head = cof.HeadTrainStep(X, Ppre, Wa, ba, Wt, bt, labels, grads, flags=flags, keep_prob=0.2, seed=42,
                                 offset=ctr, dxatt_rank1=True)

    def step():
        _, _, state['pws'] = cof.pose_head_fwd(X, W1, b1, W2, b2, workspace=state['pws'], out=(Ppre, Pl))
        _, dPl = cof.pose_l2_loss_fwd_bwd(Pl, lbl, valid)
        if rank1:
            head.run()
            cof.pose_head_bwd(X, W1, W2, Ppre, dPl, None, dX=dX, accumulate_dX=True, workspace=state['pws'],
                              ws_from_fwd=True, ext_rank1=(dZ, wa_flat))
            return
        logits, att, zs, ab, _, state['aws'] = cof.attn_pool_fwd(X, Ppre, Wa, ba, Wt, bt, flags=flags,
                                                                 keep_prob=0.2, seed=42, offset=ctr,
                                                                 workspace=state.get('aws'))
        _, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels)
        dXs, dXatt, *_ = cof.attn_pool_bwd(X, Ppre, Wa, ba, Wt, bt, att, zs, ab, G, flags=flags, keep_prob=0.2,
                                           seed=42, offset=ctr, workspace=state['aws'])
        cof.pose_head_bwd(X, W1, W2, Ppre, dPl, dXatt, dX=dXs, accumulate_dX=True, workspace=state['pws'],
                          ws_from_fwd=True)

    info = {'workload': 'cfg003 pose-regularised attention head fwd+bwd (pose head 2048->768->16 + M=1 '
                        'pooling + pose L2 + softmax-xent); per-GPU batch {} x {}x{}x{} {}, K={}, dropout '
                        'keep=0.2'.format(N, H, H, C, dtype, K),
            'bound': 'mfma', 'dtype': dtype, 'N': N,
            'flops_per_image': 3 * (2.0 * P * C * Cp + 2.0 * P * Cp * J)}      # fwd + 2x bwd
    return step, info


def build_posebwd(cof, dev, N=32, H=14, dtype='bf16', accumulate=False):
    """the pose head's backward call alone (dPpre rows pass, column sums, dW1 split-K + reduce, dX product) --
    with accumulate=False the dX product runs with beta = 0, the form a plain library GEMM is timed in
    (tools/hipblaslt_ref.py): for kernel-by-kernel comparisons under rocprofv3."""
    C, Cp, J, P = 2048, 768, 16, H * H
    td = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(42)
    X = _features(N, P, C, td, dev)
    W1 = (torch.randn(C, Cp, generator=g) / C ** 0.5).to(dev); b1 = torch.zeros(Cp, device=dev)
    W2 = (torch.randn(Cp, J, generator=g) / Cp ** 0.5).to(dev); b2 = torch.zeros(J, device=dev)
    dPl = torch.randn(N, P, J, generator=g).to(dev)
    Ppre, Pl, ws = cof.pose_head_fwd(X, W1, b1, W2, b2)
    dX = torch.zeros_like(X)

    def step():
        cof.pose_head_bwd(X, W1, W2, Ppre, dPl, None, dX=dX, accumulate_dX=accumulate, workspace=ws)
    info = {'workload': 'pose head backward call alone (beta = {}); per-GPU batch {} x {}x{}x{} {}'.format(
                int(accumulate), N, H, H, C, dtype),
            'bound': 'mfma', 'dtype': dtype, 'N': N, 'flops_per_image': 2 * (2.0 * P * C * Cp)}
    return step, info


def build_perclass(cof, dev, N=32, H=14, K=51, dtype='bf16', rotate=0, weight_images=True):
    C, P = 2048, H * H
    td = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(42)
    per_set = 2 * N * P * C * (2 if dtype == 'bf16' else 4)
    R = _n_sets(per_set, rotate)
    X = _features(N, P, C, td, dev)
    Wa = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(K, device=dev)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
    labels = torch.randint(0, K, (N,), generator=g).to(dev)
    flags = cof.attn_flags(False, False, True)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    pg = (torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt), torch.empty_like(bt))
    sts = []
    for r in range(R):            # one host call per step, one bound step per (X, dX) set
        Xr = X if r == 0 else _features(N, P, C, td, dev, seed=42 + r)
        # the padded / concatenated bf16 operand images of the weights live in the step's workspace and are kept
        # current at weight-update time (APA_FLAG_WEIGHT_IMAGES: deploy.MomentumSGD.attach_weight_images rewrites them
        # in the optimiser's own launch); here the weights do not change between steps
        sts.append(cof.HeadTrainStep(Xr, Xr, Wa, ba, Wt, bt, labels, (torch.empty_like(Xr), None) + pg, flags=flags,
                                     keep_prob=0.2, seed=42, offset=ctr, weight_images=weight_images,
                                     share_with=sts[0] if sts else None))
    esz = X.element_size()
    info = {'workload': 'per-class bottom-up maps (M=K, HMDB-51 shape when K=51) attention head fwd+bwd; '
                        'per-GPU batch {} x {}x{}x{} {}, K={}, dropout keep=0.2'.format(N, H, H, C, dtype, K) +
                        ('; weight images kept by the optimiser launch' if weight_images else '') + _rot_note(R, per_set),
            # K = 51: 0.25 GFLOP/img against 3*P*C*s bytes -> HBM-bound; K = 393: MFMA-bound
            'bound': 'hbm' if K <= 128 else 'mfma', 'dtype': dtype, 'N': N, 'rotate': R,
            'flops_per_image': 3 * (2 * 2.0 * P * C * K),              # Z and T products, fwd + 2x bwd
            'bytes_per_image': 3.0 * P * C * esz}
    return _round_robin([st.run for st in sts]), info


def _bucket_and_optimizer(dev, params, regularized, shadows=None):
    """flat gradient bucket + the momentum-SGD of the shipped configs (src/train.py:90-94; lr small enough that the
    synthetic weights stay put over thousands of timed updates, WEIGHT_DECAY as in experiments/*.yaml)"""
    from attentionalpoolingaction_amd import deploy
    bucket = deploy.GradientBucket({n: p.shape for n, p in params.items()}, dev)
    opt = deploy.MomentumSGD(params, bucket, lr=1e-7, momentum=0.9, weight_decay=5e-4, regularized=regularized,
                             bf16_shadows=shadows)
    return bucket, opt


def build_update(cof, dev, which='cfg003', prepared=True, iter_size=1, N=32, H=14, K=None, rotate=0):
    """One parameter UPDATE of a training loop, i.e. what a driver-timed `step` line leaves out (VERDICT r05 Weak #5):
    ITER_SIZE head steps (src/train.py:529-566) + the fused momentum-SGD launch.

      prepared=True   the bf16 operand copies the steps read (cfg 003: W1 shadow, W2^T image; per-class head: the
                      padded / concatenated weight images) are rewritten by the OPTIMISER'S launch
                      (apa_momentum_sgd_step_shadow / _images) -- the shipped arrangement (deploy.FusedHeadStep)
      prepared=False  the steps convert / build them from the fp32 weights in every call, the optimiser launch is the
                      plain apa_momentum_sgd_step -- the arrangement of rounds 1-3
    Both arms do the same arithmetic on the same buffers; the difference is where the preparation runs."""
    C, P = 2048, H * H
    td = torch.bfloat16
    g = torch.Generator().manual_seed(42)
    per_set = 2 * N * P * C * 2
    R = _n_sets(per_set, rotate)
    X = _features(N, P, C, td, dev)
    flags = cof.attn_flags(False, False, True)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    new = lambda t: torch.empty_like(t)
    if which == 'cfg003':
        K = K or 393
        Cp, J = 768, 16
        W1 = (torch.randn(C, Cp, generator=g) / C ** 0.5).to(dev); b1 = torch.zeros(Cp, device=dev)
        W2 = (torch.randn(Cp, J, generator=g) / Cp ** 0.5).to(dev); b2 = torch.zeros(J, device=dev)
        Wa = (torch.randn(Cp, 1, generator=g) / Cp ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
        Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
        labels = torch.randint(0, K, (N,), generator=g).to(dev)
        lbl = torch.rand(N, P, J, generator=g).to(dev)
        valid = (torch.rand(N, J, generator=g) > 0.3).to(dev)
        params = {'pose_w1': W1, 'pose_b1': b1, 'pose_w2': W2, 'pose_b2': b2, 'att_weights': Wa, 'att_biases': ba,
                  'td_weights': Wt, 'td_biases': bt}
        w1_bf16 = torch.empty(W1.shape, dtype=torch.bfloat16, device=dev) if prepared else None
        bucket, opt = _bucket_and_optimizer(dev, params, ['pose_w1', 'pose_w2', 'att_weights', 'td_weights'],
                                            {'pose_w1': w1_bf16} if prepared else None)
        w2t = None
        if prepared:
            w2t = cof.pose_w2t_image(W2)
            opt.add_image('pose_w2', cof.pose_w2t_image_map(w2t, W2), owner=w2t,
                          refresh=lambda: w2t[:J, :Cp].copy_(W2.t()))
        v = bucket.views
        sts = []
        for r in range(R):
            Xr = X if r == 0 else _features(N, P, C, td, dev, seed=42 + r)
            sts.append(cof.PoseAttnTrainStep(
                Xr, (W1, b1, W2, b2, Wa, ba, Wt, bt), labels, lbl, valid,
                (new(Xr), v['pose_w1'], v['pose_b1'], v['pose_w2'], v['pose_b2'], v['att_weights'], v['att_biases'],
                 v['td_weights'], v['td_biases']), flags=flags, keep_prob=0.2, seed=42, offset=ctr, w1_bf16=w1_bf16,
                w2t_bf16=w2t, share_with=sts[0] if sts else None))
        name = 'cfg003 head step (one host call)'
    else:
        K = K or 51
        Wa = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(K, device=dev)
        Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
        labels = torch.randint(0, K, (N,), generator=g).to(dev)
        params = {'att_weights': Wa, 'att_biases': ba, 'td_weights': Wt, 'td_biases': bt}
        bucket, opt = _bucket_and_optimizer(dev, params, ['att_weights', 'td_weights'])
        v = bucket.views
        sts = []
        for r in range(R):
            Xr = X if r == 0 else _features(N, P, C, td, dev, seed=42 + r)
            sts.append(cof.HeadTrainStep(Xr, Xr, Wa, ba, Wt, bt, labels,
                                         (new(Xr), None, v['att_weights'], v['att_biases'], v['td_weights'],
                                          v['td_biases']), flags=flags, keep_prob=0.2, seed=42, offset=ctr,
                                         weight_images=prepared, share_with=sts[0] if sts else None))
        if prepared:       # the R bound steps share ONE workspace, hence one set of images
            opt.attach_weight_images(sts[0], {'Wa': 'att_weights', 'ba': 'att_biases', 'Wt': 'td_weights',
                                              'bt': 'td_biases'})
        name = 'HMDB-51 per-class head step (K={}, one host call)'.format(K)
    rr = _round_robin([st.run for st in sts])
    scale = 1.0 / iter_size

    def update():
        for _ in range(iter_size):
            rr()
        opt.step(grad_scale=scale)
    info = {'workload': 'one UPDATE = {} x {} + the fused momentum-SGD launch ({}); per-GPU batch {} x {}x{}x{} bf16'
                        .format(iter_size, name,
                                'operand copies rewritten by the optimiser launch' if prepared else
                                'plain launch; the steps prepare their bf16 operands themselves',
                                N, H, H, C) + _rot_note(R, per_set),
            'dtype': 'bf16', 'N': N * iter_size, 'rotate': R, 'iter_size': iter_size, 'prepared': bool(prepared)}
    return update, info


def run_update_pair(cof, dev, which, min_ms=50.0, repeats=5, rotate=0):
    """the four driver-timed numbers VERDICT r05 asks for: us per update with the preparation in the optimiser's
    launch vs in every step, at TRAIN.ITER_SIZE 1 and 2"""
    out = {}
    for it in (1, 2):
        row = {}
        for prepared in (True, False):
            fn, info = build_update(cof, dev, which, prepared=prepared, iter_size=it, rotate=rotate)
            sec, reps = timed(fn, 50, 5, min_ms=min_ms, repeats=repeats)
            row['prepared_by_optimizer_us' if prepared else 'prepared_per_step_us'] = round(sec * 1e6, 2)
            row['repeats'] = reps
            row.setdefault('workload', info['workload'] if prepared else None)
            del fn
            torch.cuda.empty_cache()
        row['saving_us_per_update'] = round(row['prepared_per_step_us'] - row['prepared_by_optimizer_us'], 2)
        out['iter_size_%d' % it] = row
    return out


def build_rank1(cof, dev, N=32, H=14, K=51, dtype='bf16', rotate=0):
    """class-agnostic bottom-up map (M = 1, the shipped HMDB / MPII attention configs) on bf16 features:
    the headline op, one host call per step."""
    C, P = 2048, H * H
    td = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(42)
    X = _features(N, P, C, td, dev)
    Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
    labels = torch.randint(0, K, (N,), generator=g).to(dev)
    flags = cof.attn_flags(False, False, True)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    per_set = 2 * N * P * C * X.element_size()
    R = _n_sets(per_set, rotate)
    pg = (torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt), torch.empty_like(bt))
    sts = []
    for r in range(R):
        Xr = X if r == 0 else _features(N, P, C, td, dev, seed=42 + r)
        sts.append(cof.HeadTrainStep(Xr, Xr, Wa, ba, Wt, bt, labels, (torch.empty_like(Xr), None) + pg, flags=flags,
                                     keep_prob=0.2, seed=42, offset=ctr, share_with=sts[0] if sts else None))
    info = {'workload': 'class-agnostic map (M=1) attention head fwd + softmax-xent + bwd; per-GPU batch {} x {}x{}x{} '
                        '{}, K={}, dropout keep=0.2'.format(N, H, H, C, dtype, K) + _rot_note(R, per_set),
            'bound': 'hbm', 'dtype': dtype, 'N': N, 'rotate': R, 'flops_per_image': 0.0,
            'bytes_per_image': 3.0 * P * C * X.element_size()}
    return _round_robin([st.run for st in sts]), info


def build_eval002(cof, dev, N=32, H=14, K=393, dtype='f32', rotate=0):
    C, P = 2048, H * H
    td = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(42)
    X = _features(N, P, C, td, dev)
    Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
    per_set = N * P * C * X.element_size()              # forward only: X is all that streams
    R = _n_sets(per_set, rotate)
    evs = []
    for r in range(R):
        Xr = X if r == 0 else _features(N, P, C, td, dev, seed=42 + r)
        evs.append(cof.HeadEvalStep(Xr, Xr, Wa, ba, Wt, bt, workspace=evs[0].workspace if evs else None))
    info = {'workload': 'cfg002 eval step (attn-pool forward + softmax + argmax, one call); per-GPU batch '
                        '{} x {}x{}x{} {}, K={}'.format(N, H, H, C, dtype, K) + _rot_note(R, per_set).replace('X/dX', 'X'),
            'bound': 'hbm', 'dtype': dtype, 'N': N, 'rotate': R, 'bytes_per_image': 1.0 * P * C * X.element_size()}
    return _round_robin([ev.run for ev in evs]), info


def timed(fn, steps, warmup, min_ms=50.0, repeats=5):
    """median over >= `repeats` timed loops of `steps` steps each (>= min_ms of device time in total)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    per, total = [], 0.0
    while len(per) < repeats or total < min_ms * 1e-3:
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per.append(dt / steps)
        total += dt
        if len(per) >= 200:
            break
    per.sort()
    return per[len(per) // 2], len(per)


def report(info, sec, repeats):
    N = info['N']
    out = {'workload': info['workload'], 'images_per_sec': round(N / sec, 1), 'ms_per_step': round(sec * 1e3, 5),
           'repeats': repeats, 'dtype': info['dtype'], 'rotate': info.get('rotate', 1)}
    if info['bound'] == 'mfma':
        tflops = N * info['flops_per_image'] / sec / 1e12
        out['roofline'] = {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': PEAK[info['dtype']],
                           'unit': 'TFLOP/s', 'frac': round(tflops / PEAK[info['dtype']], 4),
                           'algorithmic_gflop_per_image': round(info['flops_per_image'] / 1e9, 3),
                           'note': 'whole step (all kernels) against the dense MFMA peak'}
    else:
        gbs = N * info['bytes_per_image'] / sec / 1e9
        out['roofline'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': round(gbs / HBM_PEAK_GBS, 4),
                           'algorithmic_bytes_per_image': info['bytes_per_image'],
                           'note': 'whole step (all kernels) against the HBM peak'}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cfg003', choices=['cfg003', 'perclass', 'eval002', 'rank1', 'posebwd', 'posebwd_acc', 'update003',
                                                         'update_perclass'])
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--hw', type=int, default=14)
    ap.add_argument('--classes', type=int, default=None)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32'])
    ap.add_argument('--per-op', action='store_true',
                    help='cfg003: drive the step as separate calls (pose head fwd, pose loss, attention train step, '
                         'pose head bwd) instead of the one-call apa_pose_attn_train_step')
    ap.add_argument('--no-rank1', action='store_true',
                    help='cfg003: materialise the [N,P,768] attention-branch gradient between the two backward calls')
    ap.add_argument('--no-weight-images', action='store_true',
                    help='perclass: rebuild the padded bf16 weight images in every step (the round-4 form)')
    ap.add_argument('--rotate', type=int, default=0,
                    help='number of (X, dX) buffer sets cycled through; 0 = enough for 1.5 x the Infinity Cache')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    args = ap.parse_args()
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    dev = torch.device('cuda:0')
    if args.workload in ('update003', 'update_perclass'):
        print(json.dumps(run_update_pair(cof, dev, 'cfg003' if args.workload == 'update003' else 'perclass',
                                         rotate=args.rotate)))
        return
    if args.workload == 'cfg003':
        step, info = build_cfg003(cof, dev, args.batch, args.hw, args.classes or 393, args.dtype or 'bf16',
                                  rank1=not args.no_rank1, one_call=not args.per_op, rotate=args.rotate)
    elif args.workload in ('posebwd', 'posebwd_acc'):
        step, info = build_posebwd(cof, dev, args.batch, args.hw, args.dtype or 'bf16',
                                   accumulate=args.workload == 'posebwd_acc')
    elif args.workload == 'rank1':
        step, info = build_rank1(cof, dev, args.batch, args.hw, args.classes or 51, args.dtype or 'bf16',
                                 rotate=args.rotate)
    elif args.workload == 'eval002':
        step, info = build_eval002(cof, dev, args.batch, args.hw, args.classes or 393, args.dtype or 'f32',
                                   rotate=args.rotate)
    else:
        step, info = build_perclass(cof, dev, args.batch, args.hw, args.classes or 51, args.dtype or 'bf16',
                                    rotate=args.rotate, weight_images=not args.no_weight_images)
    sec, reps = timed(step, args.steps, args.warmup)
    print(json.dumps(report(info, sec, reps)))


if __name__ == '__main__':
    main()
