"""GPU parity of the dense (MFMA) paths through the C ABI:
  * PoseLogits head forward/backward (nets_factory.py:147-160)
  * cfg 003 end to end: pose head -> attention from pose_pre_logits -> pose L2 + action loss
  * per-class bottom-up maps, M == K (nets_factory.py:257), id / relu / softmax, dropout
fp32 features run on the exact f32 MFMA (tight tolerance); bf16 features on the bf16 MFMA
(tolerance set by 8-bit significands)."""
import numpy as np
import pytest
import torch

from oracle import attn_pool_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _pose_problem(N, H, C, Cp, J, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).to(dtype)
    W1 = torch.randn(C, Cp, generator=g) / C ** 0.5
    b1 = torch.randn(Cp, generator=g) * 0.1
    W2 = torch.randn(Cp, J, generator=g) / Cp ** 0.5
    b2 = torch.randn(J, generator=g) * 0.1
    return X, W1, b1, W2, b2, g


@pytest.mark.parametrize('N,H,C,Cp,J', [(3, 7, 512, 768, 16), (2, 14, 2048, 768, 16), (1, 5, 256, 200, 7)])
def test_pose_head_fp32_forward_backward(gpu, N, H, C, Cp, J):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    X, W1, b1, W2, b2, g = _pose_problem(N, H, C, Cp, J, seed=N + H)
    dPl = torch.randn(N, H, H, J, generator=g)
    dExt = torch.randn(N, H, H, Cp, generator=g)
    Xr, W1r, b1r, W2r, b2r = (t.double().requires_grad_(True) for t in (X, W1, b1, W2, b2))
    pre, pl = orc.pose_logits_head(Xr, W1r, b1r, W2r, b2r)
    ((pl * dPl.double()).sum() + (pre * dExt.double()).sum()).backward()

    d = lambda t: t.to(gpu).contiguous()
    Ppre, Pl, ws = cof.pose_head_fwd(d(X), d(W1), d(b1), d(W2), d(b2))
    assert _rel(Ppre, pre) < 2e-5 and _rel(Pl, pl) < 2e-5
    dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(d(X), d(W1), d(W2), Ppre, d(dPl), d(dExt), workspace=ws)
    for name, got, want in (('dX', dX, Xr.grad), ('dW1', dW1, W1r.grad), ('db1', db1, b1r.grad),
                            ('dW2', dW2, W2r.grad), ('db2', db2, b2r.grad)):
        assert _rel(got, want) < 5e-5, name
    # accumulate mode adds onto an existing dX; dPl-only and ext-only calls are both legal
    base = torch.randn_like(dX)
    dX2, *_ = cof.pose_head_bwd(d(X), d(W1), d(W2), Ppre, d(dPl), d(dExt), dX=base.clone(), accumulate_dX=True)
    assert _rel(dX2 - base, Xr.grad) < 1e-4
    dXa, _, _, dW2a, _ = cof.pose_head_bwd(d(X), d(W1), d(W2), Ppre, None, d(dExt))
    assert float(dW2a.abs().max()) == 0.0
    dXb, *_ = cof.pose_head_bwd(d(X), d(W1), d(W2), Ppre, d(dPl), None)
    assert _rel(dXa + dXb, Xr.grad) < 1e-4


def test_pose_head_bf16(gpu):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C, Cp, J = 4, 14, 2048, 768, 16
    X, W1, b1, W2, b2, g = _pose_problem(N, H, C, Cp, J, seed=9, dtype=torch.bfloat16)
    dPl = torch.randn(N, H, H, J, generator=g)
    # reference: same bf16-rounded operands (X, W1 -> bf16 inside the kernel), float64 accumulation
    W1q = W1.bfloat16().double()
    Xr = X.double().requires_grad_(True)
    W1r = W1q.clone().requires_grad_(True)
    W2r = W2.double().requires_grad_(True)
    pre, pl = orc.pose_logits_head(Xr, W1r, b1.double(), W2r, b2.double())
    (pl * dPl.double()).sum().backward()
    d = lambda t: t.to(gpu).contiguous()
    Ppre, Pl, ws = cof.pose_head_fwd(d(X), d(W1), d(b1), d(W2), d(b2))
    assert Ppre.dtype == torch.bfloat16
    assert _rel(Ppre, pre) < 2 ** -7            # bf16 output rounding
    assert _rel(Pl, pl) < 1e-2                  # Ppre (bf16) and W2 (bf16) feed the second product
    dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(d(X), d(W1), d(W2), Ppre, d(dPl), None, workspace=ws)
    assert _rel(dW2, W2r.grad) < 2e-2 and _rel(dW1, W1r.grad) < 2e-2 and _rel(dX, Xr.grad) < 2e-2


def test_cfg003_pose_regularised_attention_end_to_end(gpu):
    """cfg 003: bottom-up map from pose_pre_logits (768 ch), top-down from conv5, loss = pose L2 +
    softmax-xent; every parameter/input gradient against autograd of the oracle."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C, Cp, J, K = 3, 7, 2048, 768, 16, 393
    X, W1, b1, W2, b2, g = _pose_problem(N, H, C, Cp, J, seed=33)
    Wa = torch.randn(Cp, 1, generator=g) / Cp ** 0.5
    ba = torch.randn(1, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    pose_lbl = torch.rand(N, H, H, J, generator=g)
    valid = torch.rand(N, J, generator=g) > 0.3

    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, W1r, b1r, W2r, b2r, War, bar, Wtr, btr = map(leaf, (X, W1, b1, W2, b2, Wa, ba, Wt, bt))
    pre, pl = orc.pose_logits_head(Xr, W1r, b1r, W2r, b2r)
    lg, _ = orc.attentional_pooling(Xr, pre, pl, [War], [bar], [Wtr], [btr],
                                    orc.AttnFlags(single_layer_att=False))
    total = sum(orc.gen_losses(labels, lg, 'softmax-xentropy', K, 1.0, pose_lbl.double(), pl, 'l2', valid, 1.0))
    total.backward()

    d = lambda t: t.to(gpu).contiguous()
    Xd, W1d, b1d, W2d, b2d, Wad, bad, Wtd, btd = map(d, (X, W1, b1, W2, b2, Wa, ba, Wt, bt))
    Ppre, Pl, pws = cof.pose_head_fwd(Xd, W1d, b1d, W2d, b2d)
    logits, att, zs, ab, _, ws = cof.attn_pool_fwd(Xd, Ppre, Wad, bad, Wtd, btd)
    lossx, G, _, pred = cof.softmax_xent_fwd_bwd(logits, d(labels), want_pred=True)
    lossp, dPl = cof.pose_l2_loss_fwd_bwd(Pl, d(pose_lbl), d(valid))
    dX, dXatt, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Ppre, Wad, bad, Wtd, btd, att, zs, ab, G, workspace=ws)
    dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xd, W1d, W2d, Ppre, dPl, dXatt, dX=dX, accumulate_dX=True,
                                               workspace=pws)
    assert abs(float(lossx[0] + lossp[0]) - float(total)) < 2e-5 * float(total)
    assert torch.equal(pred.cpu(), lg.argmax(1))
    for name, got, want in (('dX', dX, Xr.grad), ('dW1', dW1, W1r.grad), ('db1', db1, b1r.grad),
                            ('dW2', dW2, W2r.grad), ('db2', db2, b2r.grad), ('dWa', dWa, War.grad),
                            ('dba', dba, bar.grad), ('dWt', dWt, Wtr.grad), ('dbt', dbt, btr.grad)):
        assert _rel(got, want) < 1e-4, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_cfg003_one_call_attention_step_equals_the_separate_calls(gpu, dtype):
    """cfg 003 as tools/bench_dense.py drives it: pose_head_fwd into caller-owned buffers, HeadTrainStep
    bound to Ppre as its attention input with the attention-branch gradient in rank-1 form, then
    pose_head_bwd(ext_rank1).  Every result must be bit-identical to the per-op sequence (same kernels
    apart from the fused logits-reduce + cross-entropy launch, whose reduction trees are the same)."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C, Cp, J, K = 4, 7, 2048, 768, 16, 393
    P = H * H
    X, W1, b1, W2, b2, g = _pose_problem(N, H, C, Cp, J, seed=71, dtype=dtype)
    Wa = torch.randn(Cp, 1, generator=g) / Cp ** 0.5
    ba = torch.randn(1, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    pose_lbl = torch.rand(N, H, H, J, generator=g)
    valid = torch.rand(N, J, generator=g) > 0.3
    d = lambda t: t.to(gpu).contiguous()
    Xd, W1d, b1d, W2d, b2d, Wad, bad, Wtd, btd, lab = map(d, (X.view(N, P, C), W1, b1, W2, b2, Wa, ba, Wt, bt, labels))
    flags = cof.attn_flags(False, False, True)
    kw = dict(flags=flags, keep_prob=0.5, seed=9, offset=4)

    # per-op sequence
    Ppre, Pl, pws = cof.pose_head_fwd(Xd, W1d, b1d, W2d, b2d)
    logits, att, zs, ab, _, ws = cof.attn_pool_fwd(Xd, Ppre, Wad, bad, Wtd, btd, **kw)
    lossx, G, _, _ = cof.softmax_xent_fwd_bwd(logits, lab)
    _, dPl = cof.pose_l2_loss_fwd_bwd(Pl, d(pose_lbl), d(valid))
    dX, dZ, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Ppre, Wad, bad, Wtd, btd, att, zs, ab, G, workspace=ws,
                                                   dxatt_rank1=True, **kw)
    want = cof.pose_head_bwd(Xd, W1d, W2d, Ppre, dPl, None, dX=dX, accumulate_dX=True, workspace=pws,
                             ext_rank1=(dZ, Wad.view(-1)))

    # the bench's form
    Ppre2 = torch.empty((N, P, Cp), dtype=dtype, device=gpu)
    Pl2 = torch.empty((N, P, J), dtype=torch.float32, device=gpu)
    grads = (torch.empty_like(Xd), torch.empty((N * P,), dtype=torch.float32, device=gpu), torch.empty_like(Wad),
             torch.empty_like(bad), torch.empty_like(Wtd), torch.empty_like(btd))
    head = cof.HeadTrainStep(Xd, Ppre2, Wad, bad, Wtd, btd, lab, grads, dxatt_rank1=True, **kw)
    _, _, pws2 = cof.pose_head_fwd(Xd, W1d, b1d, W2d, b2d, out=(Ppre2, Pl2))
    _, dPl2 = cof.pose_l2_loss_fwd_bwd(Pl2, d(pose_lbl), d(valid))
    head.run()
    got = cof.pose_head_bwd(Xd, W1d, W2d, Ppre2, dPl2, None, dX=grads[0], accumulate_dX=True, workspace=pws2,
                            ws_from_fwd=True, ext_rank1=(grads[1], Wad.view(-1)))
    assert torch.equal(Ppre2, Ppre) and torch.equal(Pl2, Pl)
    assert torch.equal(head.logits, logits) and torch.equal(head.loss, lossx) and torch.equal(head.G, G)
    for name, a, b in (('dZ', grads[1], dZ), ('dWa', grads[2], dWa), ('dba', grads[3], dba), ('dWt', grads[4], dWt),
                       ('dbt', grads[5], dbt)):
        assert torch.equal(a, b), name
    for name, a, b in zip(('dX', 'dW1', 'db1', 'dW2', 'db2'), got, want):
        assert torch.equal(a, b), name
    with pytest.raises(cof.ApaError):
        cof.HeadTrainStep(Xd, Xd, Wtd[:, :1].contiguous(), bad, Wtd, btd, lab, grads, dxatt_rank1=True)


@pytest.mark.parametrize('dtype,N,H,softmax,relu,Cp,K', [(torch.bfloat16, 4, 7, False, False, 768, 393),
                                                          (torch.bfloat16, 32, 14, False, False, 768, 393),
                                                          (torch.bfloat16, 3, 5, True, False, 768, 393),
                                                          (torch.bfloat16, 2, 15, False, True, 768, 393),
                                                          (torch.bfloat16, 5, 9, False, False, 256, 51),
                                                          (torch.bfloat16, 1, 13, False, False, 1024, 10),
                                                          (torch.bfloat16, 7, 6, True, False, 512, 500),
                                                          (torch.float32, 3, 7, False, False, 768, 393)])
def test_cfg003_whole_step_in_one_call_matches_the_per_op_sequence(gpu, dtype, N, H, softmax, relu, Cp, K):
    """apa_pose_attn_train_step (cof.PoseAttnTrainStep): the cfg 003 head step of one sess.run -- PoseLogits head ->
    attention from pose_pre_logits -> dropout + pooling -> pose L2 + softmax cross-entropy -> all gradients
    (nets_factory.py:147-160,247-328, loss.py:11-80) -- as ONE host call, against the per-op sequence.
    bf16 features take the fused kernels (Pl product + attention logits + pose-loss gradient in one launch; dWa / dba
    on the pose head's backward rows pass; pose loss and dropout counter finished by its column-sum launch): what is
    computed by the same kernel on the same operands must be BIT-identical (Ppre, Pl, dPl), the attention logits are
    summed in a different order (1e-6), everything downstream of them follows at fp32 / bf16-storage round-off.
    fp32 features run the four calls inside the entry point: bit-identical throughout."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    C, J = 2048, 16
    P = H * H
    X, W1, b1, W2, b2, g = _pose_problem(N, H, C, Cp, J, seed=100 * N + H, dtype=dtype)
    Wa = torch.randn(Cp, 1, generator=g) / Cp ** 0.5
    ba = torch.randn(1, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    pose_lbl = torch.rand(N, P, J, generator=g)
    valid = torch.rand(N, J, generator=g) > 0.3
    d = lambda t: t.to(gpu).contiguous()
    Xd, W1d, b1d, W2d, b2d, Wad, bad, Wtd, btd, lab = map(d, (X.view(N, P, C), W1, b1, W2, b2, Wa, ba, Wt, bt, labels))
    lbl_d, valid_d = d(pose_lbl), d(valid)
    flags = cof.attn_flags(softmax, relu, True)
    ctr_a = torch.full((1,), 7, dtype=torch.int64, device=gpu)
    ctr_b = ctr_a.clone()
    wts = dict(action_wt=1.3, pose_wt=0.7, grad_scale=0.5)

    # ---- per-op sequence (device-side dropout counter, like the one-call form)
    kw = dict(flags=flags, keep_prob=0.5, seed=9, offset=ctr_a)
    Ppre, Pl, pws = cof.pose_head_fwd(Xd, W1d, b1d, W2d, b2d)
    lossp, dPl = cof.pose_l2_loss_fwd_bwd(Pl, lbl_d, valid_d, wt=wts['pose_wt'], grad_scale=wts['grad_scale'])
    logits, att, zs, ab, _, ws = cof.attn_pool_fwd(Xd, Ppre, Wad, bad, Wtd, btd, **kw)
    lossx, G, _, _ = cof.softmax_xent_fwd_bwd(logits, lab, wt=wts['action_wt'], grad_scale=wts['grad_scale'])
    dX, dZ, dWa, dba, dWt, dbt = cof.attn_pool_bwd(Xd, Ppre, Wad, bad, Wtd, btd, att, zs, ab, G, workspace=ws,
                                                   dxatt_rank1=True, **kw)
    dXf, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xd, W1d, W2d, Ppre, dPl, None, dX=dX, accumulate_dX=True,
                                                workspace=pws, ext_rank1=(dZ, Wad.view(-1)))
    torch.cuda.synchronize()
    assert int(ctr_a) == 8

    # ---- one call (with and without the caller-maintained bf16 copy of W1)
    # (the bf16 operand images a caller may keep: W1 as is, W2 transposed -- same values, so nothing may change)
    variants = ((None, None), (W1d.to(torch.bfloat16), None), (W1d.to(torch.bfloat16), cof.pose_w2t_image(W2d))) \
        if dtype == torch.bfloat16 else ((None, None),)
    for shadow, w2t in variants:
        ctr_b.fill_(7)
        e = lambda t: torch.full_like(t, float('nan'))
        grads = (e(Xd), e(W1d), e(b1d), e(W2d), e(b2d), e(Wad), e(bad), e(Wtd), e(btd))
        st = cof.PoseAttnTrainStep(Xd, (W1d, b1d, W2d, b2d, Wad, bad, Wtd, btd), lab, lbl_d, valid_d, grads,
                                   flags=flags, keep_prob=0.5, seed=9, offset=ctr_b, w1_bf16=shadow, w2t_bf16=w2t,
                                   **wts)
        st.run()
        torch.cuda.synchronize()
        assert int(ctr_b) == 8                                  # the dropout counter advanced exactly once
        exact = dtype == torch.float32
        assert torch.equal(st.Ppre, Ppre) and torch.equal(st.Pl, Pl) and torch.equal(st.dPl.view_as(dPl), dPl)
        tol = 0.0 if exact else 2e-6
        assert _rel(st.att, att) <= tol and _rel(st.loss_pose, lossp) <= (0.0 if exact else 1e-6)
        got = dict(logits=st.logits, loss=st.loss_action, G=st.G, dZ=st.dZ, dWa=grads[5], dba=grads[6],
                   dWt=grads[7], dbt=grads[8], dW1=grads[1], db1=grads[2], dW2=grads[3], db2=grads[4])
        want = dict(logits=logits, loss=lossx, G=G, dZ=dZ, dWa=dWa, dba=dba, dWt=dWt, dbt=dbt, dW1=dW1, db1=db1,
                    dW2=dW2, db2=db2)
        for k_ in got:
            if exact:
                assert torch.equal(got[k_].view_as(want[k_]), want[k_]), k_
            elif k_ == 'dba' and softmax:      # a spatial softmax is shift invariant: exactly 0, round-off on both sides
                assert float(got[k_].abs().max()) < 1e-5 * float(want['dZ'].abs().sum())
            else:
                assert _rel(got[k_], want[k_]) < 2e-4, (k_, _rel(got[k_], want[k_]))
        if exact:
            assert torch.equal(grads[0], dXf)
        else:                                                   # bf16 storage: an ulp of the output format
            assert _rel(grads[0].float(), dXf.float()) < 1.0 / 128
            assert not torch.isnan(grads[0].float()).any()


@pytest.mark.parametrize('K,dtype,N', [(130, torch.bfloat16, 3), (51, torch.bfloat16, 3), (51, torch.bfloat16, 70),
                                         (10, torch.bfloat16, 33), (70, torch.float32, 3)])
def test_per_class_one_call_train_step_equals_the_separate_calls(gpu, K, dtype, N):
    """Per-class maps through apa_attn_head_train_step: the backward half reuses what the forward half left in
    the workspace (APA_FLAG_WS_FROM_FWD: padded bf16 weights and the materialised dropout(X) of the generic
    path for K > 64, the prepared slab and the keep bits of the fused path for K <= 64) -- every output must
    equal the per-op sequence bit for bit."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    P, C = 49, 512
    g = torch.Generator().manual_seed(K)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(dtype).to(gpu)
    Wa = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu); ba = (torch.randn(K, generator=g) * 0.1).to(gpu)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu); bt = (torch.randn(K, generator=g) * 0.1).to(gpu)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    kw = dict(flags=cof.attn_flags(False, True, True), keep_prob=0.5, seed=3, offset=2)

    def grads():
        return (torch.empty_like(X), None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt),
                torch.empty_like(bt))
    ga, gb = grads(), grads()
    st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, ga, **kw)
    st.run()
    logits, att, Ts, _, _, ws = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, **kw)
    loss, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels)
    cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, att, Ts, None, G, workspace=ws, out=gb, **kw)
    torch.cuda.synchronize()
    # (K <= 64, the fused HMDB-51 path: inside the one call the row's cross-entropy is computed by the activation pass
    # itself -- one launch less -- with softmax_xent_kernel's arithmetic to the letter, and the batch mean by the dW
    # reduce launch's tail in that kernel's summation order: still bit for bit)
    for a, b, name in ((st.logits, logits, 'logits'), (st.att, att, 'att'), (st.loss, loss, 'loss'), (st.G, G, 'G'),
                       (ga[0], gb[0], 'dX'), (ga[2], gb[2], 'dWa'), (ga[3], gb[3], 'dba'), (ga[4], gb[4], 'dWt'),
                       (ga[5], gb[5], 'dbt')):
        assert torch.equal(a, b), name
    assert bool(torch.isfinite(ga[0].float()).all()) and float(ga[4].abs().max()) > 0


@pytest.mark.parametrize('softmax', [False, True])
@pytest.mark.parametrize('N,H,C', [(3, 7, 2048), (5, 6, 768)])
def test_per_class_keep_bits_prepared_by_the_previous_step(gpu, N, H, C, softmax):
    """K <= 64, bf16, caller-kept weight images: the one-call train step runs NO preparation launch -- its last launch
    leaves the NEXT step's dropout decisions behind, tagged (seed, offset + 1), and the forward kernel believes the
    map only if the tag matches its own (seed, offset); else (first step on a workspace, a jump of the offset) it hashes
    its rows itself.  Four steps on a device-side counter (fallback, then three believed maps), one jump of the
    counter (fallback again), a separately called forward / backward pair in between (its preparation launch rewrites
    the map and says so): every output bit-identical to the step that prepares everything per call."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    K, P = 51, H * H
    g = torch.Generator().manual_seed(77 + N)
    Xs = [torch.relu(torch.randn(N, P, C, generator=g)).to(torch.bfloat16).to(gpu) for _ in range(2)]
    mk = lambda *s_: (torch.randn(*s_, generator=g) / (C ** 0.5 if len(s_) == 2 else 10.0)).to(gpu)
    Wa, ba, Wt, bt = mk(C, K), mk(K), mk(C, K), mk(K)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    flags = cof.attn_flags(softmax, False, True)
    ca = torch.full((1,), 7, dtype=torch.int64, device=gpu)
    cb = ca.clone()

    def make(weight_images, ctr, X):
        grads = (torch.full_like(X, float('nan')), None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt),
                 torch.empty_like(bt))
        return cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, flags=flags, keep_prob=0.2, seed=5, offset=ctr,
                                 weight_images=weight_images), grads
    a, ga = make(True, ca, Xs[0])
    b, gb = make(False, cb, Xs[0])
    assert a._args[-5] & cof.APA_FLAG_WEIGHT_IMAGES

    def same(tag):
        a.run(); b.run()
        torch.cuda.synchronize()
        assert int(ca) == int(cb)
        for name in ('logits', 'att', 'zsave', 'loss', 'G'):
            assert torch.equal(getattr(a, name), getattr(b, name)), (tag, name)
        for x, y in zip(ga, gb):
            if x is not None:
                assert torch.equal(x, y) and not torch.isnan(x.float()).any(), tag
    for i in range(4):
        if i == 2:                      # another feature map, same workspace: the mask does not depend on X
            a.rebind(X=Xs[1]); b.rebind(X=Xs[1])
        same('step %d' % i)
    assert int(ca) == 11
    ca.fill_(40); cb.fill_(40)          # the counter jumps: the prepared map is for offset 11
    same('after the jump')
    same('after the jump + 1')
    # a separately called forward on the SAME workspace (its preparation launch writes the map of offset 99) ...
    lg, att, zs, _, _, _ = cof.attn_pool_fwd(Xs[0], Xs[0], Wa, ba, Wt, bt, flags=flags | cof.APA_FLAG_WEIGHT_IMAGES,
                                             keep_prob=0.2, seed=5, offset=99, workspace=a.workspace)
    lg2, _, _, _, _, _ = cof.attn_pool_fwd(Xs[0], Xs[0], Wa, ba, Wt, bt, flags=flags, keep_prob=0.2, seed=5, offset=99)
    assert torch.equal(lg, lg2)
    same('after a foreign forward call')      # ... is noticed: the tag no longer says offset 42
    same('and once more')


def test_per_class_tagged_step_replays_from_a_hipgraph(gpu):
    """The stateful per-class step stays hipGraph-capturable: no host decision depends on the tag -- the forward kernel
    checks it on the device against the HBM step counter -- so ONE captured step (caller-kept weight images, keep bits
    from the previous replay's last launch) replayed four times draws four fresh masks and equals, bit for bit, the
    eager stateless step at offsets 0, 1, 2, 3."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C, K = 4, 7, 2048, 51
    P = H * H
    g = torch.Generator().manual_seed(5)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(torch.bfloat16).to(gpu)
    mk = lambda *s_: (torch.randn(*s_, generator=g) / (C ** 0.5 if len(s_) == 2 else 10.0)).to(gpu)
    Wa, ba, Wt, bt = mk(C, K), mk(K), mk(C, K), mk(K)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    flags = cof.attn_flags(False, False, True)

    def make(weight_images, offset):
        grads = (torch.empty_like(X), None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt),
                 torch.empty_like(bt))
        return cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, flags=flags, keep_prob=0.2, seed=3, offset=offset,
                                 weight_images=weight_images), grads
    ref = []
    for i in range(4):
        st, gr = make(False, i)
        st.run()
        torch.cuda.synchronize()
        ref.append((st.logits.clone(), st.loss.clone(), gr[0].clone(), gr[4].clone()))
    assert not torch.equal(ref[0][0], ref[1][0])
    ctr = torch.zeros(1, dtype=torch.int64, device=gpu)
    a, ga = make(True, ctr)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a.run(stream=side.cuda_stream)                   # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ctr.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a.run()
    for i in range(4):
        graph.replay()
        torch.cuda.synchronize()
        assert int(ctr) == i + 1
        for got, want in zip((a.logits, a.loss, ga[0], ga[4]), ref[i]):
            assert torch.equal(got, want), 'replay %d' % i


@pytest.mark.parametrize('K,dtype,train', [(51, torch.bfloat16, True), (51, torch.bfloat16, False),
                                           (130, torch.bfloat16, True), (70, torch.float32, True)])
def test_per_class_weight_images_kept_by_the_optimizer_launch(gpu, K, dtype, train):
    """APA_FLAG_WEIGHT_IMAGES (include/apa.h): the padded / concatenated operand images of the per-class head are
    built once in the step's workspace (apa_per_class_weight_images) and rewritten by the OPTIMISER'S launch
    (apa_momentum_sgd_step_images) -- the per-step preparation launch then writes only the keep bits.  Every output of
    the one-call step must be BIT-identical to the step that rebuilds the images from the fp32 weights in every call:
    right after construction, and again after two optimiser updates."""
    from attentionalpoolingaction_amd import deploy
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C = 3, 7, 2048
    P = H * H
    g = torch.Generator().manual_seed(1000 + K)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(dtype).to(gpu)
    shapes = {'att_weights': (C, K), 'att_biases': (K,), 'td_weights': (C, K), 'td_biases': (K,)}
    params = {n: (torch.randn(s_, generator=g) / (C ** 0.5 if len(s_) == 2 else 10.0)).to(gpu) for n, s_ in shapes.items()}
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    flags = cof.attn_flags(False, False, train)

    def make(weight_images):
        grads = (torch.full_like(X, float('nan')), None) + tuple(torch.full_like(params[n], float('nan')) for n in shapes)
        st = cof.HeadTrainStep(X, X, params['att_weights'], params['att_biases'], params['td_weights'],
                               params['td_biases'], labels, grads, flags=flags, keep_prob=0.5, seed=11, offset=3,
                               weight_images=weight_images)
        return st, grads
    a, ga = make(True)
    b, gb = make(False)
    assert a.weight_image_maps and not b.weight_image_maps
    roles = {r for r, _ in a.weight_image_maps}
    assert roles >= {'Wa', 'ba', 'Wt'} and (K > 64 or dtype != torch.bfloat16 or 'bt' in roles)
    bucket = deploy.GradientBucket(shapes, gpu)
    opt = deploy.MomentumSGD(params, bucket, lr=0.05, momentum=0.9, weight_decay=5e-4,
                             regularized=['att_weights', 'td_weights'])
    opt.attach_weight_images(a, {'Wa': 'att_weights', 'ba': 'att_biases', 'Wt': 'td_weights', 'bt': 'td_biases'})
    assert opt.images and not opt._img_refresh

    def same():
        a.run(); b.run()
        torch.cuda.synchronize()
        for name in ('logits', 'att', 'zsave', 'loss', 'G'):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        for x, y in zip(ga, gb):
            if x is not None:
                assert torch.equal(x, y) and not torch.isnan(x.float()).any()
    same()
    for _ in range(2):
        bucket.flat.copy_(torch.randn(bucket.flat.numel(), generator=g).to(gpu))
        opt.step()
        same()
    # evaluation on the same workspace: no preparation launch at all, same logits
    ev_a = cof.HeadEvalStep(X, X, params['att_weights'], params['att_biases'], params['td_weights'], params['td_biases'],
                            flags=cof.APA_FLAG_WEIGHT_IMAGES, workspace=a.workspace)
    ev_b = cof.HeadEvalStep(X, X, params['att_weights'], params['att_biases'], params['td_weights'], params['td_biases'])
    ev_a.run(); ev_b.run()
    torch.cuda.synchronize()
    assert torch.equal(ev_a.logits, ev_b.logits) and torch.equal(ev_a.pred, ev_b.pred)
    # the images are laid out for 16-byte aligned features: an odd address is refused, not silently re-prepared
    if dtype == torch.bfloat16:
        Xo = torch.empty(N * P * C + 8, dtype=dtype, device=gpu)[1:1 + N * P * C].view(N, P, C)
        with pytest.raises(cof.ApaError, match='16-byte'):
            cof.attn_pool_fwd(Xo, Xo, params['att_weights'], params['att_biases'], params['td_weights'],
                              params['td_biases'], flags=cof.APA_FLAG_WEIGHT_IMAGES, workspace=a.workspace)


@pytest.mark.parametrize('K', [51, 130])
def test_per_class_device_side_dropout_counter_is_advanced_by_the_backward_call(gpu, K):
    """APA_FLAG_RNG_DEVICE (include/apa.h): `offset` is the address of a step counter in HBM that apa_attn_pool_bwd
    advances when it is done -- also on the per-class paths (fused K <= 64 and generic), per-op and one-call: every
    step draws a fresh mask, and the counter never moves before the last kernel that keys its mask with it."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, P, C = 3, 49, 512
    g = torch.Generator().manual_seed(K + 1)
    X = torch.relu(torch.randn(N, P, C, generator=g)).to(torch.bfloat16).to(gpu)
    Wa = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu); ba = (torch.randn(K, generator=g) * 0.1).to(gpu)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(gpu); bt = (torch.randn(K, generator=g) * 0.1).to(gpu)
    labels = torch.randint(0, K, (N,), generator=g).to(gpu)
    flags = cof.attn_flags(False, False, True)
    ctr = torch.full((1,), 5, dtype=torch.int64, device=gpu)
    kw = dict(flags=flags, keep_prob=0.5, seed=3, offset=ctr)
    logits, att, Ts, _, _, ws = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, **kw)
    _, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels)
    dX1 = cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, att, Ts, None, G, workspace=ws, **kw)[0].clone()
    torch.cuda.synchronize()
    assert int(ctr) == 6
    # the same step keyed by value with offset 5 gives the same gradient: the counter was read as 5 throughout
    kv = dict(flags=flags, keep_prob=0.5, seed=3, offset=5)
    l2, a2, T2, _, _, ws2 = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, **kv)
    _, G2, _, _ = cof.softmax_xent_fwd_bwd(l2, labels)
    dX2 = cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, a2, T2, None, G2, workspace=ws2, **kv)[0]
    assert torch.equal(dX1, dX2)
    # one-call steps: +1 each, and a different mask each
    grads = (torch.empty_like(X), None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt),
             torch.empty_like(bt))
    st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, **kw)
    st.run(); torch.cuda.synchronize()
    first = grads[0].clone()
    st.run(); torch.cuda.synchronize()
    assert int(ctr) == 8 and not torch.equal(first, grads[0])


def _pc_problem(N, H, C, K, seed, Ca=None, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    Ca = C if Ca is None else Ca
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).to(dtype)
    Xatt = X if Ca == C else torch.relu(torch.randn(N, H, H, Ca, generator=g)).to(dtype)
    Wa = torch.randn(Ca, K, generator=g) / Ca ** 0.5
    ba = torch.randn(K, generator=g) * 0.1
    Wt = torch.randn(C, K, generator=g) / C ** 0.5
    bt = torch.randn(K, generator=g) * 0.1
    labels = torch.randint(0, K, (N,), generator=g)
    return X, Xatt, Wa, ba, Wt, bt, labels


def _pc_run(gpu, X, Xatt, Wa, ba, Wt, bt, labels, softmax, relu, train=False, keep=0.5, seed=5, offset=2,
            want_topdown=False):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    d = lambda t: t.to(gpu).contiguous()
    Xd = d(X)
    Xad = Xd if Xatt is X else d(Xatt)
    flags = cof.attn_flags(softmax, relu, train)
    logits, att, Ts, _, td, ws = cof.attn_pool_fwd(Xd, Xad, d(Wa), d(ba), d(Wt), d(bt), flags=flags,
                                                   keep_prob=keep, seed=seed, offset=offset,
                                                   want_topdown=want_topdown)
    loss, G, _, pred = cof.softmax_xent_fwd_bwd(logits, d(labels), want_pred=True)
    grads = cof.attn_pool_bwd(Xd, Xad, d(Wa), d(ba), d(Wt), d(bt), att, Ts, None, G, flags=flags,
                              keep_prob=keep, seed=seed, offset=offset, workspace=ws)
    return logits, att, td, loss, pred, grads


@pytest.mark.parametrize('K,softmax,relu', [(51, False, False), (393, False, False), (51, True, False),
                                            (20, False, True)])
def test_per_class_maps_fp32(gpu, K, softmax, relu):
    N, H, C = 3, 6, 512
    X, Xatt, Wa, ba, Wt, bt, labels = _pc_problem(N, H, C, K, seed=K)
    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, War, bar, Wtr, btr = map(leaf, (X, Wa, ba, Wt, bt))
    flags = orc.AttnFlags(per_class=True, softmax_att=softmax, relu_att=relu)
    lg, ep = orc.attentional_pooling(Xr, None, None, [War], [bar], [Wtr], [btr], flags)
    orc.action_softmax_xent(lg, labels, K).backward()
    logits, att, td, loss, pred, (dX, dXatt, dWa, dba, dWt, dbt) = _pc_run(
        gpu, X, Xatt, Wa, ba, Wt, bt, labels, softmax, relu, want_topdown=True)
    assert dXatt is None
    assert _rel(logits, lg) < 2e-5 and float((logits.cpu().double() - lg.detach()).abs().max()) < 1e-3
    assert torch.equal(pred.cpu(), lg.argmax(1))
    assert _rel(att.view(N, H, H, K), ep['PosePrelogitsBasedAttention']) < 2e-5
    assert _rel(td.view(N, H, H, K), ep['TopDownAttention']) < 2e-5          # end-point dump
    for name, got, want in (('dX', dX, Xr.grad), ('dWa', dWa, War.grad), ('dba', dba, bar.grad),
                            ('dWt', dWt, Wtr.grad), ('dbt', dbt, btr.grad)):
        tol = 5e-5 if not (softmax and name == 'dba') else 1e-3                # d(ba) == 0 under softmax
        err = float((got.cpu().double().reshape(-1) - want.reshape(-1)).abs().max())
        assert err <= tol * max(float(want.abs().max()), 1e-30) + (1e-8 if name == 'dba' else 0), name


def test_per_class_maps_training_dropout_and_separate_attention_input(gpu):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C, K, Ca = 2, 5, 512, 51, 96
    X, Xatt, Wa, ba, Wt, bt, labels = _pc_problem(N, H, C, K, seed=77, Ca=Ca)
    keep, seed, offset = 0.5, 21, 4
    mask = cof.dropout_mask(tuple(X.shape), keep, seed, offset).cpu()
    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr, Xar, War, bar, Wtr, btr = map(leaf, (X, Xatt, Wa, ba, Wt, bt))
    flags = orc.AttnFlags(single_layer_att=False, per_class=True)
    lg, _ = orc.attentional_pooling(Xr, Xar, None, [War], [bar], [Wtr], [btr], flags, is_training=True,
                                    keep_prob=keep, dropout_mask=mask)
    orc.action_softmax_xent(lg, labels, K).backward()
    logits, att, _, loss, pred, (dX, dXatt, dWa, dba, dWt, dbt) = _pc_run(
        gpu, X, Xatt, Wa, ba, Wt, bt, labels, False, False, train=True, keep=keep, seed=seed, offset=offset)
    assert _rel(logits, lg) < 2e-5
    for name, got, want in (('dX', dX, Xr.grad), ('dXatt', dXatt, Xar.grad), ('dWa', dWa, War.grad),
                            ('dba', dba, bar.grad), ('dWt', dWt, Wtr.grad), ('dbt', dbt, btr.grad)):
        assert _rel(got, want) < 5e-5, name


def test_per_class_maps_bf16_hmdb_shape(gpu):
    """HMDB-51 config of BASELINE.json: 51 classes, bf16 features, per-class maps on the bf16 MFMA."""
    N, H, C, K = 4, 14, 2048, 51
    X, Xatt, Wa, ba, Wt, bt, labels = _pc_problem(N, H, C, K, seed=51, dtype=torch.bfloat16)
    leaf = lambda t: t.double().clone().requires_grad_(True)
    Xr = leaf(X)
    War, Wtr = leaf(Wa.bfloat16()), leaf(Wt.bfloat16())       # weights are rounded to bf16 in-kernel
    lg, _ = orc.attentional_pooling(Xr, None, None, [War], [ba.double()], [Wtr], [bt.double()],
                                    orc.AttnFlags(per_class=True))
    orc.action_softmax_xent(lg, labels, K).backward()
    logits, att, _, loss, pred, (dX, _, dWa, dba, dWt, dbt) = _pc_run(gpu, X, Xatt, Wa, ba, Wt, bt, labels, False, False)
    assert _rel(logits, lg) < 2e-3
    assert float((logits.cpu().double() - lg.detach()).abs().max()) < 5e-3
    assert _rel(dWt, Wtr.grad) < 2e-2 and _rel(dWa, War.grad) < 2e-2 and _rel(dX, Xr.grad) < 3e-2
