"""bf16 parity of the BASELINE bf16 configs against the float64 oracle, with a STATED error model
instead of flat tolerances.

What "the same inputs" means here: the feature map X is generated in bf16 (exactly representable, so
the oracle sees the identical numbers); every PARAMETER stays fp32 and reaches the oracle unrounded --
the kernels' own bf16 rounding of weights and intermediates is part of the error being tested, not
something the oracle is allowed to share.

Error model (u = 2^-8, the unit roundoff of bf16's 8-bit significand; fp32 accumulation in the MFMA):
  * a product  y = sum_i a_i b_i  whose operand b is rounded to bf16 inside the kernel satisfies
        |y_hat - y| <= (u + n * 2^-24) * sum_i |a_i| |b_i|          (first order; n = reduction length)
    and twice that when both operands are rounded;
  * a tensor STORED in bf16 adds u * |y| elementwise;
  * an error e in an input tensor propagates linearly: |e| . |W|.
The bounds below are computed elementwise in float64 from the oracle's own intermediates with those
three rules (matmuls of absolute values), so they scale with the data instead of being tuned constants.
For gradients, whose upstream errors pass through the loss, the direct rounding terms of the final
product are bounded the same way and the propagated part is covered by KAPPA * u * max|reference|.

These are worst-case (every rounding error at its maximum and aligned) bounds: rigorous, but loose by
sqrt(n) for long reductions.  For the LOGITS, which north_star pins at 1e-3 in fp32, two sharper checks
are made on top:
  * against a float64 oracle that rounds exactly where the kernels round (W1 -> bf16, pose_pre_logits
    stored in bf16, W2 -> bf16): the kernels' own arithmetic must agree with it to 3e-4;
  * against the unrounded oracle: <= LOGIT_TOL_BF16 = 3e-3.  That figure is what the bf16 FORMAT costs
    on this config, not kernel error: the 768-long attention dot over a bf16-stored pose_pre_logits map
    carries sigma ~ 1e-3 per pixel, and the part of it that comes from rounding W1 is the same for every
    pixel of the batch, so it does not average out in the spatial mean.  The test prints both distances.
"""
import numpy as np
import pytest
import torch

from oracle import attn_pool_oracle as orc

pytestmark = pytest.mark.gpu

U = 2.0 ** -8
KAPPA = 3.0          # propagated first-order error of a gradient tensor, in units of u * max|ref|
F32 = 2.0 ** -24
LOGIT_TOL_BF16 = 3e-3


def _bf16(t):
    """round-to-nearest-even to bf16, returned in float64"""
    return t.float().bfloat16().double()


def _viol(got, ref, bound, floor=1e-7):
    """max over elements of |got - ref| / (bound + floor): <= 1 means inside the model."""
    err = (got.detach().cpu().double().reshape(ref.shape) - ref).abs()
    return float((err / (bound + floor)).max()), float(err.max())


@pytest.mark.parametrize('N,H', [(16, 14), (32, 14), (16, 15), (32, 15)])
def test_cfg003_bf16_train_step_vs_float64_oracle(gpu, N, H):
    """BASELINE configs[2] (003_MPII_ResNet_withPoseAttention, bf16) at its real sizes: PoseLogits head
    (2048 -> 768 -> 16) -> attention from pose_pre_logits -> dropout(keep 0.2) + pooling -> pose L2 +
    softmax cross-entropy -> backward of both heads, through the C ABI, against autograd of the float64
    oracle on the same bf16 feature map (nets_factory.py:147-160,247-328, loss.py:11-80)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    C, Cp, J, K, P = 2048, 768, 16, 393, H * H
    g = torch.Generator().manual_seed(1000 * N + H)
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).bfloat16()
    W1 = torch.randn(C, Cp, generator=g) / C ** 0.5
    b1 = torch.randn(Cp, generator=g) * 0.1
    W2 = torch.randn(Cp, J, generator=g) / Cp ** 0.5
    b2 = torch.randn(J, generator=g) * 0.1
    Wa = torch.randn(Cp, 1, generator=g) / Cp ** 0.5
    ba = torch.randn(1, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    pose_lbl = torch.rand(N, H, H, J, generator=g)
    valid = torch.rand(N, J, generator=g) > 0.3
    keep, seed, offset = 0.2, 42, 5

    # ---------------- HIP path ----------------
    d = lambda t: t.to(gpu).contiguous()
    Xd, W1d, b1d, W2d, b2d, Wad, bad, Wtd, btd = map(d, (X, W1, b1, W2, b2, Wa, ba, Wt, bt))
    flags = cof.attn_flags(False, False, True)
    Ppre, Pl, pws = cof.pose_head_fwd(Xd, W1d, b1d, W2d, b2d)
    logits, att, zs, ab, _, ws = cof.attn_pool_fwd(Xd, Ppre, Wad, bad, Wtd, btd, flags=flags, keep_prob=keep,
                                                   seed=seed, offset=offset)
    lossx, G, _, pred = cof.softmax_xent_fwd_bwd(logits, d(labels), want_pred=True)
    lossp, dPl = cof.pose_l2_loss_fwd_bwd(Pl, d(pose_lbl), d(valid))
    dX, dZ, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Ppre, Wad, bad, Wtd, btd, att, zs, ab, G, flags=flags,
                                                   keep_prob=keep, seed=seed, offset=offset, workspace=ws,
                                                   dxatt_rank1=True)
    dX_pool = dX.clone()                       # the pooling op's own share, before the pose head adds to it
    dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xd, W1d, W2d, Ppre, dPl, None, dX=dX, accumulate_dX=True,
                                               workspace=pws, ext_rank1=(dZ, Wad.view(-1)))
    mask = cof.dropout_mask((N, H, H, C), keep, seed, offset).cpu()
    torch.cuda.synchronize()

    # ---------------- float64 oracle: bf16 INPUTS, unrounded parameters ----------------
    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, W1r, b1r, W2r, b2r, War, bar, Wtr, btr = map(leaf, (X, W1, b1, W2, b2, Wa, ba, Wt, bt))
    pre, pl = orc.pose_logits_head(Xr, W1r, b1r, W2r, b2r)
    pre.retain_grad()
    pl.retain_grad()
    lg, ep = orc.attentional_pooling(Xr, pre, pl, [War], [bar], [Wtr], [btr], orc.AttnFlags(single_layer_att=False),
                                     is_training=True, keep_prob=keep, dropout_mask=mask)
    ep['PosePrelogitsBasedAttention'].retain_grad()
    l_pose, l_act = orc.gen_losses(labels, lg, 'softmax-xentropy', K, 1.0, pose_lbl.double(), pl, 'l2', valid, 1.0)
    (l_pose + l_act).backward()

    # ---------------- forward error model ----------------
    with torch.no_grad():
        Xa, W1a = Xr.abs().reshape(N * P, C), W1r.abs()
        prea = pre.abs().reshape(N * P, Cp)
        B_pre = ((U + C * F32) * (Xa @ W1a) + U * prea)                       # W1 rounded + bf16 store
        B_pl = B_pre @ W2r.abs() + (U + Cp * F32) * (prea @ W2r.abs())        # + W2 rounded
        B_att = B_pre @ War.abs() + Cp * F32 * (prea @ War.abs())             # fp32 GEMV on the bf16 map
        Xt = (Xr * mask.double() / keep).reshape(N, P, C)
        T = Xt @ Wtr + btr                                                    # [N,P,K], exact inputs
        B_logit = torch.einsum('np,npk->nk', B_att.reshape(N, P), T.abs()) / P \
            + 4e-6 * torch.einsum('np,npk->nk', ep['PosePrelogitsBasedAttention'].abs().reshape(N, P), T.abs()) / P
    v_pre, e_pre = _viol(Ppre.float(), pre.detach().reshape(N, P, Cp), B_pre.reshape(N, P, Cp))
    v_pl, e_pl = _viol(Pl, pl.detach().reshape(N, P, J), B_pl.reshape(N, P, J))
    v_att, e_att = _viol(att, ep['PosePrelogitsBasedAttention'].detach().reshape(N, P, 1), B_att.reshape(N, P, 1))
    v_lg, e_lg = _viol(logits, lg.detach(), B_logit)
    print('N={} H={}: forward error / model bound: Ppre {:.3f} Pl {:.3f} att {:.3f} logits {:.3f}; max abs err '
          'Ppre {:.2e} Pl {:.2e} att {:.2e} logits {:.2e}'.format(N, H, v_pre, v_pl, v_att, v_lg, e_pre, e_pl, e_att, e_lg))
    assert v_pre <= 1.0 and v_pl <= 1.0 and v_att <= 1.0 and v_lg <= 1.0
    # the same graph with the kernels' rounding points (W1, W2 -> bf16; pose_pre_logits stored in bf16)
    with torch.no_grad():
        pre_q = _bf16(torch.relu(Xr.reshape(N * P, C) @ _bf16(W1r) + b1r)).reshape(N, H, H, Cp)
        lg_q, _ = orc.attentional_pooling(Xr, pre_q, None, [War], [bar], [Wtr], [btr],
                                          orc.AttnFlags(single_layer_att=False), is_training=True,
                                          keep_prob=keep, dropout_mask=mask)
    e_kernel = float((logits.cpu().double() - lg_q).abs().max())
    e_format = float((lg_q - lg.detach()).abs().max())
    print('   logits: kernel vs rounding-aware oracle {:.2e}; rounding-aware vs unrounded oracle (the bf16 format) '
          '{:.2e}; kernel vs unrounded {:.2e}'.format(e_kernel, e_format, e_lg))
    assert e_kernel < 3e-4
    assert e_lg < LOGIT_TOL_BF16
    # argmax: bit-exact wherever the reference's own top-2 margin exceeds that uncertainty
    top2 = lg.detach().topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    sure = margin > 2.0 * LOGIT_TOL_BF16
    assert int(sure.sum()) >= int(0.75 * N), 'test problem too ambiguous: {} / {} rows decidable'.format(int(sure.sum()), N)
    assert torch.equal(pred.cpu()[sure], lg.detach().argmax(1)[sure])
    assert torch.equal(pred.cpu(), logits.argmax(1).cpu())            # and always the argmax of its own logits
    # losses: softmax-xent is 2-Lipschitz in the max-norm of a row; pose L2 to first order in B_pl
    assert abs(float(lossx[0]) - float(l_act)) <= 2.0 * e_lg + 1e-6
    dpl = (pl.detach() - pose_lbl.double()).abs().reshape(N, P, J) * valid.double().reshape(N, 1, J)
    b_pose = float(((2 * dpl * B_pl.reshape(N, P, J) + B_pl.reshape(N, P, J) ** 2)).sum()) * 0.5 / (N * N * P)
    assert abs(float(lossp[0]) - float(l_pose)) <= b_pose + 1e-7

    # ---------------- backward ----------------
    got = {'dW1': dW1, 'db1': db1, 'dW2': dW2, 'db2': db2, 'dWa': dWa, 'dba': dba, 'dWt': dWt, 'dbt': dbt,
           'dX': dX.float().view(N, H, H, C)}
    ref = {'dW1': W1r.grad, 'db1': b1r.grad, 'dW2': W2r.grad, 'db2': b2r.grad, 'dWa': War.grad, 'dba': bar.grad,
           'dWt': Wtr.grad, 'dbt': btr.grad, 'dX': Xr.grad}

    def relmax(a, b):
        return float((a.detach().cpu().double().reshape(b.shape) - b).abs().max() / b.abs().max())

    # (1) the backward KERNELS, checked in isolation: float64 evaluation of the backward formulas on the
    # tensors the kernels actually consumed (their own bf16 pose_pre_logits -- hence identical ReLU
    # gates --, dPl, dZ), so only the operand roundings of these products separate the two:
    #   dPpre = [Ppre > 0] * (dPl . W2^T + dZ (x) wa)   -> bf16 for the MFMA (u), W2 / W1 rounded (u each)
    with torch.no_grad():
        pre_g = Ppre.cpu().double().reshape(N * P, Cp)
        dpl_g = dPl.cpu().double().reshape(N * P, J)
        gpre = (dpl_g @ W2r.t() + dZ.cpu().double().reshape(N * P, 1) * War.reshape(1, Cp)) * (pre_g > 0)
        exp = {'dW1': Xr.reshape(N * P, C).t() @ gpre, 'db1': gpre.sum(0), 'dW2': pre_g.t() @ dpl_g,
               'db2': dpl_g.sum(0),
               'dX': (dX_pool.cpu().double().reshape(N * P, C) + gpre @ W1r.t()).reshape(N, H, H, C)}
    kern = {k: relmax(got[k], exp[k]) for k in exp}
    print('   backward kernels vs float64 formulas on their own inputs (units of u): ' +
          ', '.join('{} {:.2f}u'.format(k, v / U) for k, v in kern.items()))
    for k, v in kern.items():
        assert v <= 3.0 * U, (k, v / U)
    # (2) against autograd of the unrounded oracle: what the bf16 FORMAT does to the gradients on this
    # problem, on top of (1).  Tensors reached only through the attention / classifier path see the att
    # error alone (observed <= 1.3u, budget 3u).  The pose-head gradients are driven by dPl = c (Pl - lbl),
    # a difference whose forward error (B_pl, ~3u of |Pl - lbl|) is carried back through two bf16-rounded
    # layers and a bf16-stored dPpre; dW1 and dX additionally sum thousands of signed terms, so a
    # correlated error does not cancel the way the terms themselves do: observed 9-20u, budget 32u, while
    # (1) shows the kernels contribute < 1u of it.  Reported, and held to those budgets.
    GRAD_U = {'dW1': 32, 'db1': 8, 'dW2': 3, 'db2': 3, 'dX': 32, 'dWa': 3, 'dba': 3, 'dWt': 3, 'dbt': 3}
    full = {k: relmax(got[k], ref[k]) for k in got}
    print('   backward vs autograd of the unrounded oracle (units of u): ' +
          ', '.join('{} {:.2f}u'.format(k, v / U) for k, v in full.items()))
    with torch.no_grad():
        y1 = (Xr.reshape(N * P, C) @ W1r + b1r)
        flipped = float(((y1 > 0) != (pre_g > 0)).double().mean())
    print('   ReLU gates that differ between the bf16 kernels and the unrounded oracle: {:.3%}'.format(flipped))
    assert flipped < 0.005
    for k, v in full.items():
        assert v <= GRAD_U[k] * U, (k, v / U)


@pytest.mark.parametrize('N', [8, 32])
def test_hmdb51_per_class_bf16_vs_float64_oracle(gpu, N):
    """BASELINE configs[4] (HMDB-51, 51 classes, bf16, per-class bottom-up maps M = K on the bf16 MFMA,
    nets_factory.py:257): logits within the north_star 1e-3 of the float64 oracle on the same bf16
    features with UNROUNDED fp32 parameters, inside the error model, argmax bit-exact."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    H, C, K, P = 14, 2048, 51, 196
    g = torch.Generator().manual_seed(51 + N)
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).bfloat16()
    Wa = torch.randn(C, K, generator=g) / C ** 0.5
    ba = torch.randn(K, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    d = lambda t: t.to(gpu).contiguous()
    Xd = d(X)
    logits, att, Ts, _, _, ws = cof.attn_pool_fwd(Xd, Xd, d(Wa), d(ba), d(Wt), d(bt))
    loss, G, _, pred = cof.softmax_xent_fwd_bwd(logits, d(labels), want_pred=True)
    dX, _, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Xd, d(Wa), d(ba), d(Wt), d(bt), att, Ts, None, G, workspace=ws)
    torch.cuda.synchronize()

    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, War, bar, Wtr, btr = map(leaf, (X, Wa, ba, Wt, bt))
    lg, ep = orc.attentional_pooling(Xr, None, None, [War], [bar], [Wtr], [btr], orc.AttnFlags(per_class=True))
    orc.action_softmax_xent(lg, labels, K).backward()
    with torch.no_grad():
        Xa = Xr.abs().reshape(N, P, C)
        Z = ep['PosePrelogitsBasedAttention'].reshape(N, P, K)
        T = ep['TopDownAttention'].reshape(N, P, K)
        B_Z = (U + C * F32) * (Xa @ War.abs())              # Wa rounded to bf16
        B_T = (U + C * F32) * (Xa @ Wtr.abs())              # Wt rounded to bf16
        B_logit = ((B_Z * T.abs() + Z.abs() * B_T).sum(1)) / P
    v_att, e_att = _viol(att, Z, B_Z)
    v_lg, e_lg = _viol(logits, lg.detach(), B_logit)
    print('N={}: per-class bf16 error / model bound: att {:.3f} logits {:.3f}; max abs err att {:.2e} logits {:.2e}; '
          'bound on logits max {:.2e}'.format(N, v_att, v_lg, e_att, e_lg, float(B_logit.max())))
    assert v_att <= 1.0 and v_lg <= 1.0
    with torch.no_grad():     # the same graph with the kernel's rounding points: Wa, Wt -> bf16
        lg_q, _ = orc.attentional_pooling(Xr, None, None, [_bf16(War)], [bar], [_bf16(Wtr)], [btr],
                                          orc.AttnFlags(per_class=True))
    e_kernel = float((logits.cpu().double() - lg_q).abs().max())
    print('   logits: kernel vs rounding-aware oracle {:.2e}; rounding-aware vs unrounded oracle (the bf16 format) '
          '{:.2e}'.format(e_kernel, float((lg_q - lg.detach()).abs().max())))
    assert e_kernel < 3e-4
    assert e_lg < LOGIT_TOL_BF16
    top2 = lg.detach().topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2.0 * LOGIT_TOL_BF16
    assert int(sure.sum()) >= int(0.75 * N)
    assert torch.equal(pred.cpu()[sure], lg.detach().argmax(1)[sure])
    assert torch.equal(pred.cpu(), logits.argmax(1).cpu())
    # gradients: both operands of each product are bf16 (weights rounded in-kernel, G-side tensors rounded
    # when staged): direct term 2u |A|.|B| plus the propagated part
    for name, got, ref in (('dWt', dWt, Wtr.grad), ('dWa', dWa, War.grad), ('dbt', dbt, btr.grad),
                           ('dba', dba, bar.grad), ('dX', dX.float().view(N, H, H, C), Xr.grad)):
        rel = float((got.detach().cpu().double().reshape(ref.shape) - ref).abs().max() / ref.abs().max())
        print('   {} rel err {:.2e} (= {:.2f} u)'.format(name, rel, rel / U))
        assert rel <= (2.0 + KAPPA) * U, name


@pytest.mark.parametrize('N,K,softmax', [(32, 51, False), (5, 51, True), (3, 64, False), (4, 10, False)])
def test_per_class_small_k_bf16_training_fused_kernels(gpu, N, K, softmax):
    """The HBM-bound per-class path for K <= 64 (apa_pc_fused.hip: Z | T in one pass over X with the dropout
    applied on the way into LDS, dWt | dWa in one pass, dX in one launch) in TRAINING mode, keep = 0.2:
    against the float64 oracle fed the kernel's own mask -- logits vs the rounding-aware oracle (Wa, Wt
    -> bf16), gradients in units of u -- and the one-call train step (which lets the backward reuse the
    operands the forward prepared, APA_FLAG_WS_FROM_FWD) bit-identical to the three separate calls."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    H, C, P = 14, 2048, 196
    g = torch.Generator().manual_seed(7 * N + K)
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).bfloat16()
    Wa = torch.randn(C, K, generator=g) / C ** 0.5
    ba = torch.randn(K, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    keep, seed, offset = 0.2, 11, 3
    d = lambda t: t.to(gpu).contiguous()
    Xd, Wad, bad, Wtd, btd, lab = d(X), d(Wa), d(ba), d(Wt), d(bt), d(labels)
    flags = cof.attn_flags(softmax, False, True)
    logits, att, Ts, _, _, ws = cof.attn_pool_fwd(Xd, Xd, Wad, bad, Wtd, btd, flags=flags, keep_prob=keep, seed=seed,
                                                  offset=offset)
    loss, G, _, pred = cof.softmax_xent_fwd_bwd(logits, lab, want_pred=True)
    dX, _, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Xd, Wad, bad, Wtd, btd, att, Ts, None, G, flags=flags,
                                                  keep_prob=keep, seed=seed, offset=offset)   # fresh workspace
    mask = cof.dropout_mask((N, H, H, C), keep, seed, offset).cpu()
    # the one-call step on the same inputs
    grads = (torch.empty_like(Xd), None, torch.empty_like(Wad), torch.empty_like(bad), torch.empty_like(Wtd),
             torch.empty_like(btd))
    st = cof.HeadTrainStep(Xd, Xd, Wad, bad, Wtd, btd, lab, grads, flags=flags, keep_prob=keep, seed=seed,
                           offset=offset)
    st.run()
    torch.cuda.synchronize()
    assert torch.equal(st.logits, logits) and torch.equal(st.G, G)
    for a, b, name in zip(grads, (dX, None, dWa, dba, dWt, dbt), ('dX', '', 'dWa', 'dba', 'dWt', 'dbt')):
        if a is not None:
            assert torch.equal(a, b), name

    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, War, bar, Wtr, btr = map(leaf, (X, Wa, ba, Wt, bt))
    oflags = orc.AttnFlags(per_class=True, softmax_att=softmax)
    lg, ep = orc.attentional_pooling(Xr, None, None, [War], [bar], [Wtr], [btr], oflags, is_training=True,
                                     keep_prob=keep, dropout_mask=mask)
    orc.action_softmax_xent(lg, labels, K).backward()
    with torch.no_grad():
        lg_q, _ = orc.attentional_pooling(Xr, None, None, [_bf16(War)], [bar], [_bf16(Wtr)], [btr], oflags,
                                          is_training=True, keep_prob=keep, dropout_mask=mask)
    e_kernel = float((logits.cpu().double() - lg_q).abs().max())
    e_full = float((logits.cpu().double() - lg.detach()).abs().max())
    print('N={} K={}: logits kernel vs rounding-aware oracle {:.2e}, vs unrounded {:.2e}'.format(N, K, e_kernel, e_full))
    assert e_kernel < 3e-4 and e_full < 2 * LOGIT_TOL_BF16        # keep = 0.2 scales T (and its error) by 5
    assert torch.equal(pred.cpu(), logits.argmax(1).cpu())
    for name, got, ref in (('dWt', dWt, Wtr.grad), ('dWa', dWa, War.grad), ('dbt', dbt, btr.grad),
                           ('dba', dba, bar.grad), ('dX', dX.float().view(N, H, H, C), Xr.grad)):
        scale = max(float(ref.abs().max()), 1e-30)
        rel = float((got.detach().cpu().double().reshape(ref.shape) - ref).abs().max()) / scale
        print('   {} rel err {:.2e} (= {:.2f} u)'.format(name, rel, rel / U))
        if softmax and name == 'dba':
            assert float(got.abs().max()) < 1e-6        # d(ba) == 0 under the spatial softmax
            continue
        assert rel <= (2.0 + KAPPA) * U, name


@pytest.mark.parametrize('N,H,C,K,relu,train', [(3, 7, 512, 5, True, True),      # R = 147: ragged row blocks, relu
                                                (2, 6, 256, 64, False, False),   # K = 64, no dropout (kernel<false>)
                                                (5, 14, 512, 51, True, True),    # images straddle 32- and 128-row blocks
                                                (9, 7, 256, 2, True, False),     # K < 4: logits by the finish kernel
                                                (1, 5, 256, 3, False, True),     # P = 25 < 32: the separate passes
                                                (40, 6, 256, 17, False, True)])  # 5 images per 128-row block
def test_per_class_small_k_folded_activation_passes(gpu, N, H, C, K, relu, train):
    """Round 4: for identity / relu attention the per-class K <= 64 path has no activation launches left -- the
    forward product's epilogue writes A = f(Z) and per-block partial rows of sum_p A * T, the backward dX kernel forms
    [dT | dZ] in registers (apa_pc_fused.hip).  Shapes that exercise what the HMDB-51 benchmark shape does not: row
    counts that are not multiples of the 32 / 128-row blocks, several images per block, relu, evaluation mode, K < 4
    (no folded cross-entropy) and P < 32 (falls back to the separate passes).  Against the float64 oracle fed the
    kernel's own mask, in units of u; one-call step bit-identical to the per-op sequence."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    g = torch.Generator().manual_seed(1000 * N + 10 * K + H)
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).bfloat16()
    Wa = torch.randn(C, K, generator=g) / C ** 0.5
    ba = torch.randn(K, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    keep, seed, offset = (0.5, 13, 6) if train else (1.0, 0, 0)
    d = lambda t: t.to(gpu).contiguous()
    Xd, Wad, bad, Wtd, btd, lab = d(X), d(Wa), d(ba), d(Wt), d(bt), d(labels)
    flags = cof.attn_flags(False, relu, train)
    kw = dict(flags=flags, keep_prob=keep, seed=seed, offset=offset)
    logits, att, Ts, _, _, ws = cof.attn_pool_fwd(Xd, Xd, Wad, bad, Wtd, btd, **kw)
    loss, G, _, pred = cof.softmax_xent_fwd_bwd(logits, lab, want_pred=True)
    dX, _, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Xd, Wad, bad, Wtd, btd, att, Ts, None, G, **kw)
    grads = (torch.empty_like(Xd), None, torch.empty_like(Wad), torch.empty_like(bad), torch.empty_like(Wtd),
             torch.empty_like(btd))
    st = cof.HeadTrainStep(Xd, Xd, Wad, bad, Wtd, btd, lab, grads, **kw)
    st.run()
    torch.cuda.synchronize()
    assert torch.equal(st.logits, logits) and torch.equal(st.G, G) and torch.equal(st.att, att)
    assert torch.equal(st.loss, loss)
    for a, b, name in zip(grads, (dX, None, dWa, dba, dWt, dbt), ('dX', '', 'dWa', 'dba', 'dWt', 'dbt')):
        if a is not None:
            assert torch.equal(a, b), name

    mask = cof.dropout_mask((N, H, H, C), keep, seed, offset).cpu() if train else None
    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, War, bar, Wtr, btr = map(leaf, (X, Wa, ba, Wt, bt))
    oflags = orc.AttnFlags(per_class=True, relu_att=relu)
    okw = dict(is_training=True, keep_prob=keep, dropout_mask=mask) if train else {}
    lg, ep = orc.attentional_pooling(Xr, None, None, [War], [bar], [Wtr], [btr], oflags, **okw)
    orc.action_softmax_xent(lg, labels, K).backward()
    with torch.no_grad():
        lg_q, ep_q = orc.attentional_pooling(Xr, None, None, [_bf16(War)], [bar], [_bf16(Wtr)], [btr], oflags, **okw)
    assert float((logits.cpu().double() - lg_q).abs().max()) < 3e-4
    assert float((logits.cpu().double() - lg.detach()).abs().max()) < 2 * LOGIT_TOL_BF16
    a_q = ep_q['PosePrelogitsBasedAttention'].reshape(N, H * H, K)
    assert float((att.cpu().double().view(N, H * H, K) - a_q).abs().max()) < 3e-4 * max(1.0, float(a_q.abs().max()))
    assert torch.equal(pred.cpu(), logits.argmax(1).cpu())
    for name, got, ref in (('dWt', dWt, Wtr.grad), ('dWa', dWa, War.grad), ('dbt', dbt, btr.grad),
                           ('dba', dba, bar.grad), ('dX', dX.float().view(N, H, H, C), Xr.grad)):
        scale = max(float(ref.abs().max()), 1e-30)
        rel = float((got.detach().cpu().double().reshape(ref.shape) - ref).abs().max()) / scale
        assert rel <= (2.0 + KAPPA) * U, (name, rel / U)


@pytest.mark.parametrize('N,K,relu,train', [(32, 130, False, True), (32, 393, False, True), (32, 70, True, False),
                                            (33, 70, False, True)])     # 33: the last row tile is ragged
def test_per_class_large_k_one_launch_dx_at_the_benchmark_batch(gpu, N, K, relu, train):
    """Per-class maps with K > 64 at the benchmark batch (32 x 14x14x2048, bf16): the shape at which the generic
    path's dX = (dT.Wt^T)*mask/keep + dZ.Wa^T is ONE product over the concatenated contraction with the accumulators
    masked in between (gemm_bf16_wide_kernel<.., MID>, apa_gemm_bf16.hip), Z | T and dWt | dWa are twin products of one
    launch each and the cross-entropy is taken by the backward activation pass -- smaller batches fall back to the
    two-product form.  Against the float64 oracle fed the kernel's own mask, in units of u; one-call == per-op."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    H, C = 14, 2048
    g = torch.Generator().manual_seed(31 * K + N)
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).bfloat16()
    Wa = torch.randn(C, K, generator=g) / C ** 0.5
    ba = torch.randn(K, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    keep, seed, offset = (0.5, 17, 2) if train else (1.0, 0, 0)
    d = lambda t: t.to(gpu).contiguous()
    Xd, Wad, bad, Wtd, btd, lab = d(X), d(Wa), d(ba), d(Wt), d(bt), d(labels)
    kw = dict(flags=cof.attn_flags(False, relu, train), keep_prob=keep, seed=seed, offset=offset)
    logits, att, Ts, _, _, ws = cof.attn_pool_fwd(Xd, Xd, Wad, bad, Wtd, btd, **kw)
    loss, G, _, pred = cof.softmax_xent_fwd_bwd(logits, lab, want_pred=True)
    dX, _, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Xd, Wad, bad, Wtd, btd, att, Ts, None, G, **kw)
    grads = (torch.empty_like(Xd), None, torch.empty_like(Wad), torch.empty_like(bad), torch.empty_like(Wtd),
             torch.empty_like(btd))
    st = cof.HeadTrainStep(Xd, Xd, Wad, bad, Wtd, btd, lab, grads, **kw)
    st.run()
    torch.cuda.synchronize()
    assert torch.equal(st.logits, logits) and torch.equal(st.G, G) and torch.equal(st.loss, loss)
    for a, b, name in zip(grads, (dX, None, dWa, dba, dWt, dbt), ('dX', '', 'dWa', 'dba', 'dWt', 'dbt')):
        if a is not None:
            assert torch.equal(a, b), name
    mask = cof.dropout_mask((N, H, H, C), keep, seed, offset).cpu() if train else None
    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, War, bar, Wtr, btr = map(leaf, (X, Wa, ba, Wt, bt))
    oflags = orc.AttnFlags(per_class=True, relu_att=relu)
    okw = dict(is_training=True, keep_prob=keep, dropout_mask=mask) if train else {}
    lg, ep = orc.attentional_pooling(Xr, None, None, [War], [bar], [Wtr], [btr], oflags, **okw)
    orc.action_softmax_xent(lg, labels, K).backward()
    assert float((logits.cpu().double() - lg.detach()).abs().max()) < 2 * LOGIT_TOL_BF16
    # relu attention: a gate within rounding of 0 flips with the bf16 rounding of the weights (tools/fuzz_all.py skips
    # such cases; among 2.5 M map elements ~ 1 200 always do).  A flipped gate moves its whole row of dX by one of the
    # row's K terms -- those rows are compared apart, the weight gradients (sums over all rows) get a wider band.
    a_ref = ep['PosePrelogitsBasedAttention'].detach().reshape(N * H * H, K)
    flip_rows = ((att.cpu().double().view(N * H * H, K) > 0) != (a_ref > 0)).any(1) if relu else torch.zeros(N * H * H, dtype=torch.bool)
    assert int(flip_rows.sum()) < 0.25 * N * H * H
    for name, got, ref in (('dWt', dWt, Wtr.grad), ('dWa', dWa, War.grad), ('dbt', dbt, btr.grad),
                           ('dba', dba, bar.grad), ('dX', dX.float().view(N, H, H, C), Xr.grad)):
        scale = max(float(ref.abs().max()), 1e-30)
        err = (got.detach().cpu().double().reshape(ref.shape) - ref).abs()
        if name == 'dX':
            err = err.view(N * H * H, C)[~flip_rows]
        rel = float(err.max()) / scale
        print('   K={} {} rel err {:.2e} (= {:.2f} u)'.format(K, name, rel, rel / U))
        assert rel <= (2.0 + KAPPA) * U * (3.0 if (relu and name != 'dX') else 1.0), (name, rel / U)
