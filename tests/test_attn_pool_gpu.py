"""GPU parity: HIP attentional pooling (through the C ABI) vs the CPU oracle, M == 1 path.

Tolerance: BASELINE.json north_star -> 1e-3 abs fp32 on outputs, argmax bit-exact.  We hold the
fp32 kernels to a much tighter bound against the float64 oracle.
"""
import itertools

import pytest
import torch

from oracle import attn_pool_oracle as orc
from tests._synth import make_head_inputs

pytestmark = pytest.mark.gpu

ATOL_LOGITS = 1e-3        # north_star tolerance
TIGHT = 2e-5              # what fp32 kernels actually achieve vs the fp64 oracle (relative to scale)


def _oracle(inp, flags, dtype=torch.float64, train=False, keep=0.2, mask=None, G_fn=None):
    X = inp['X'].to(dtype).clone().requires_grad_(True)
    fused = inp['Xatt'] is inp['X']
    Xatt = X if fused else inp['Xatt'].to(dtype).clone().requires_grad_(True)
    Wa = inp['Wa'].to(dtype).clone().requires_grad_(True)
    ba = inp['ba'].to(dtype).clone().requires_grad_(True)
    Wt = inp['Wt'].to(dtype).clone().requires_grad_(True)
    bt = inp['bt'].to(dtype).clone().requires_grad_(True)
    logits, ep = orc.attentional_pooling(
        X, None if fused else Xatt, None, [Wa], [ba], [Wt], [bt], flags, is_training=train,
        keep_prob=keep, dropout_mask=mask)
    loss = orc.action_softmax_xent(logits, inp['labels'], logits.shape[1], 1.0)
    loss.backward()
    out = dict(logits=logits.detach(), att=ep['PosePrelogitsBasedAttention'].detach(), loss=loss.detach(),
               dX=X.grad, dWa=Wa.grad, dba=ba.grad, dWt=Wt.grad, dbt=bt.grad)
    if not fused:
        out['dXatt'] = Xatt.grad
    return out


def _run_hip(inp, dev, *, softmax=False, relu=False, train=False, keep=0.2, seed=1234, offset=7):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    X = inp['X'].to(dev).contiguous()
    fused = inp['Xatt'] is inp['X']
    Xatt = X if fused else inp['Xatt'].to(dev).contiguous()
    Wa, ba, Wt, bt = (inp[k].to(dev).contiguous() for k in ('Wa', 'ba', 'Wt', 'bt'))
    labels = inp['labels'].to(dev)
    flags = cof.attn_flags(softmax, relu, train)
    logits, att, zsave, abar, _, ws = cof.attn_pool_fwd(
        X, Xatt, Wa, ba, Wt, bt, flags=flags, keep_prob=keep, seed=seed, offset=offset)
    lossbuf, G, probs, pred = cof.softmax_xent_fwd_bwd(logits, labels, want_probs=True, want_pred=True)
    dX, dXatt, dWa, dba, dWt, dbt = cof.attn_pool_bwd(
        X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, flags=flags, keep_prob=keep, seed=seed,
        offset=offset, workspace=ws)
    torch.cuda.synchronize()
    return dict(logits=logits.cpu(), att=att.cpu(), loss=lossbuf[0].cpu(), dX=dX.cpu(),
                dXatt=None if dXatt is None else dXatt.cpu(), dWa=dWa.cpu(), dba=dba.cpu(),
                dWt=dWt.cpu(), dbt=dbt.cpu(), pred=pred.cpu(), probs=probs.cpu())


def _close(a, b, tol, what, absolute=False, atol=0.0):
    """max|a-b| <= tol * max|b| + atol  (or <= tol when absolute=True)."""
    a = a.double().reshape(-1)
    b = b.double().reshape(-1)
    scale = 1.0 if absolute else max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    assert err <= tol * scale + atol, '{}: max abs err {:.3e} > {:.1e} * scale {:.3g}'.format(
        what, err, tol, scale)


def _grads_close(got, ref, keys, tol=5e-5):
    for k in keys:
        # softmax is shift-invariant: the true d(ba) is exactly 0 there, so it needs an absolute floor
        _close(got[k].reshape(ref[k].shape), ref[k], tol, k, atol=1e-8 if k == 'dba' else 0.0)


CASES = [
    # H, C, K, softmax, relu
    (14, 2048, 393, False, False),   # cfg 002 (BASELINE shape)
    (15, 2048, 393, False, False),   # reference-native 450^2 crop
    (7, 2048, 51, False, True),
    (14, 2048, 393, True, False),    # north_star wording: spatial softmax
    (14, 1024, 51, True, True),      # BN-Inception tap (C=1024)
    (5, 512, 10, False, False),      # vgg_16/conv5 tap (C=512)
]


@pytest.mark.parametrize('H,C,K,softmax,relu', CASES)
def test_m1_fused_fp32_parity(gpu, H, C, K, softmax, relu):
    inp = make_head_inputs(N=6, H=H, W=H, C=C, K=K, seed=42 + H + K, relu_x=(C != 512))
    flags = orc.AttnFlags(single_layer_att=True, softmax_att=softmax, relu_att=relu)
    ref = _oracle(inp, flags)
    got = _run_hip(inp, gpu, softmax=softmax, relu=relu)
    _close(got['logits'], ref['logits'], ATOL_LOGITS, 'logits (north_star tol)', absolute=True)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['att'].reshape(ref['att'].shape), ref['att'], TIGHT, 'attention map')
    _close(got['loss'], ref['loss'], TIGHT, 'loss')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1)), 'argmax must be bit-exact'
    _grads_close(got, ref, ('dX', 'dWa', 'dba', 'dWt', 'dbt'))


@pytest.mark.parametrize('H,C,softmax', [(36, 256, True), (36, 256, False), (34, 1024, True)])
def test_m1_batch_one_large_map_split_count_is_clamped(gpu, H, C, softmax):
    """The shipped eval config runs batch 1; with P > 1024 pixels the per-image split count
    (target 512 blocks / N) would exceed the 256 splits m1_finalize_fwd_kernel merges -- m1_plan clamps
    it.  Both kernel families (per-pixel C = 256, channel-split C = 1024), softmax statistics included."""
    inp = make_head_inputs(N=1, H=H, W=H, C=C, K=10, seed=5 + H)
    flags = orc.AttnFlags(single_layer_att=True, softmax_att=softmax)
    ref = _oracle(inp, flags)
    got = _run_hip(inp, gpu, softmax=softmax)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['att'].reshape(ref['att'].shape), ref['att'], TIGHT, 'attention map')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))
    _grads_close(got, ref, ('dX', 'dWa', 'dba', 'dWt', 'dbt'))


@pytest.mark.parametrize('C,softmax,relu,train', [(832, False, False, False), (1536, True, False, False),
                                                  (1280, False, True, True), (2064, False, False, True),
                                                  (96, True, False, True)])
def test_m1_any_channel_count_generic_kernels(gpu, C, softmax, relu, train):
    """The head is backbone-agnostic (nets_factory.py:63-67): channel counts the register-resident kernels
    were not instantiated for -- 832 / 1536 (Inception taps), 1280, 2064 (= 2048 + 16, the concatenated
    map of _WITH_POSE_FEAT taken literally), 96 -- run on the shape-generic kernels (apa_m1_generic.hip)
    instead of returning APA_ERR_UNSUPPORTED; same oracle, same tolerances, dropout with the kernel's mask."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    inp = make_head_inputs(N=3, H=7, W=7, C=C, K=51, seed=C)
    keep, seed, offset = 0.5, 5, 2
    mask = cof.dropout_mask(tuple(inp['X'].shape), keep, seed, offset).cpu() if train else None
    flags = orc.AttnFlags(single_layer_att=True, softmax_att=softmax, relu_att=relu)
    ref = _oracle(inp, flags, train=train, keep=keep, mask=mask)
    got = _run_hip(inp, gpu, softmax=softmax, relu=relu, train=train, keep=keep, seed=seed, offset=offset)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['att'].reshape(ref['att'].shape), ref['att'], TIGHT, 'attention map')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))
    _grads_close(got, ref, ('dX', 'dWa', 'dba', 'dWt', 'dbt'))


@pytest.mark.parametrize('N,H,C,K,softmax,relu', [
    (20, 5, 256, 7, False, True),      # C < 512: fewer than 32 role-A blocks have to cover 32 rows of sn
    (3, 7, 1280, 2, False, True),      # K = 2: Kp must still cover one whole 4-wide MFMA k step
    (4, 3, 1280, 513, True, False),    # K = 1 (mod 32), more than 4 dbt columns per block
    (5, 4, 256, 33, False, False), (5, 4, 256, 34, True, False), (2, 3, 512, 1, False, False)])
def test_m1_backward_fallback_kernel_shapes_found_by_the_random_sweep(gpu, N, H, C, K, softmax, relu):
    """Shapes on which m1_bwd_small_kernel (the backward head for C < 512, K < 4 or many dbt columns per
    block) was wrong until tools/fuzz_attn_pool.py found them: its LDS rows were padded to Kp >= K only, so
    for K = 1, 2 (mod 32) the last MFMA k step read the next row; and with C / 16 < 32 blocks only the first
    C / 16 rows of every 32-row tile received sn = G . bt.  dX / dWa / dba were off by up to 300x."""
    inp = make_head_inputs(N=N, H=H, W=H, C=C, K=K, seed=N + K)
    flags = orc.AttnFlags(single_layer_att=True, softmax_att=softmax, relu_att=relu)
    ref = _oracle(inp, flags)
    got = _run_hip(inp, gpu, softmax=softmax, relu=relu)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    floor = 1e-6 * float(ref['dWt'].abs().max())
    for k in ('dX', 'dWa', 'dba', 'dWt', 'dbt'):
        _close(got[k].reshape(ref[k].shape), ref[k], 5e-5, k, atol=floor)


def test_m1_generic_kernels_separate_attention_input_bf16(gpu):
    """generic arm, cfg 003 wiring (attention map from a 200-channel tensor), bf16 features, C = 1000"""
    inp = make_head_inputs(N=2, H=6, W=6, C=1000, K=10, Ca=200, seed=3)
    inp['X'] = inp['X'].bfloat16().float()
    inp['Xatt'] = inp['Xatt'].bfloat16().float()
    flags = orc.AttnFlags(single_layer_att=False)
    ref = _oracle(inp, flags)
    bf = dict(inp)
    bf['X'], bf['Xatt'] = inp['X'].bfloat16(), inp['Xatt'].bfloat16()
    got = _run_hip(bf, gpu)
    _close(got['logits'], ref['logits'], 1e-4, 'logits')
    _close(got['dWt'].reshape(ref['dWt'].shape), ref['dWt'], 1e-4, 'dWt')
    _close(got['dX'].float().reshape(ref['dX'].shape), ref['dX'], 2.0 ** -7, 'dX (bf16 store)')


@pytest.mark.parametrize('softmax,relu', [(False, False), (True, False), (False, True)])
def test_m1_separate_attention_input_cfg003(gpu, softmax, relu):
    # cfg 003: bottom-up map from pose_pre_logits (768 channels), top-down from conv5
    inp = make_head_inputs(N=5, H=14, W=14, C=2048, K=393, Ca=768, seed=7)
    flags = orc.AttnFlags(single_layer_att=False, softmax_att=softmax, relu_att=relu)
    ref = _oracle(inp, flags)
    got = _run_hip(inp, gpu, softmax=softmax, relu=relu)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['att'].reshape(ref['att'].shape), ref['att'], TIGHT, 'attention map')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))
    _grads_close(got, ref, ('dX', 'dXatt', 'dWa', 'dba', 'dWt', 'dbt'))


@pytest.mark.parametrize('softmax', [False, True])
def test_m1_training_dropout_matches_oracle_with_same_mask(gpu, softmax):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    inp = make_head_inputs(N=4, H=14, W=14, C=2048, K=393, seed=11)
    keep, seed, offset = 0.2, 99, 3          # keep = 0.2: nets_factory.py:145 default
    mask = cof.dropout_mask(tuple(inp['X'].shape), keep, seed, offset).cpu()
    frac = mask.float().mean().item()
    assert abs(frac - keep) < 5e-3, 'dropout keep fraction {:.4f}'.format(frac)
    flags = orc.AttnFlags(single_layer_att=True, softmax_att=softmax)
    ref = _oracle(inp, flags, train=True, keep=keep, mask=mask)
    got = _run_hip(inp, gpu, softmax=softmax, train=True, keep=keep, seed=seed, offset=offset)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _grads_close(got, ref, ('dX', 'dWa', 'dba', 'dWt', 'dbt'))
    # a different (seed, offset) must give a different mask
    mask2 = cof.dropout_mask(tuple(inp['X'].shape), keep, seed, offset + 1).cpu()
    assert not torch.equal(mask, mask2)


def test_m1_bf16_features(gpu):
    inp = make_head_inputs(N=4, H=14, W=14, C=2048, K=393, seed=5, dtype=torch.bfloat16)
    ref = _oracle(inp, orc.AttnFlags())           # oracle sees the bf16-rounded X in float64
    got = _run_hip(inp, gpu)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')   # fp32 accumulation of bf16 inputs
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))
    _close(got['dWt'], ref['dWt'], 5e-5, 'dWt')
    _close(got['dWa'].reshape(ref['dWa'].shape), ref['dWa'], 5e-5, 'dWa')
    # dX is written in bf16 (8 significant bits): elementwise |err| <= 2^-8 |ref| + fp32 noise
    err = (got['dX'].double() - ref['dX']).abs()
    slack = 1e-5 * float(ref['dX'].abs().max())
    assert float((err - ref['dX'].abs() * 2 ** -8).max()) <= slack


def test_m1_reference_init_scale(gpu):
    # reference initialisation: weights ~ N(0, 1e-3), biases 0 (nets_factory.py:141,265-266,301-302)
    inp = make_head_inputs(N=3, H=15, W=15, C=2048, K=393, seed=3, w_std=1e-3, bias_std=0.0)
    ref = _oracle(inp, orc.AttnFlags())
    got = _run_hip(inp, gpu)
    _close(got['logits'], ref['logits'], 1e-6, 'logits (abs, tiny scale)', absolute=True)
    _close(got['logits'], ref['logits'], TIGHT, 'logits (relative)')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))


def test_m1_large_batch_linearity(gpu):
    """Size-independent property at the BASELINE batch: logits are linear in Wt/bt, so
    f(X; Wt1+Wt2, bt1+bt2) == f(X; Wt1, bt1) + f(X; Wt2, bt2) to rounding, and each image's
    result is independent of its batch neighbours."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    inp = make_head_inputs(N=32, H=14, W=14, C=2048, K=393, seed=21)
    X = inp['X'].to(gpu)
    Wa, ba, Wt, bt = (inp[k].to(gpu) for k in ('Wa', 'ba', 'Wt', 'bt'))
    Wt2, bt2 = torch.randn_like(Wt) * 0.02, torch.randn_like(bt) * 0.1
    l1 = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt)[0]
    l2 = cof.attn_pool_fwd(X, X, Wa, ba, Wt2, bt2)[0]
    l12 = cof.attn_pool_fwd(X, X, Wa, ba, Wt + Wt2, bt + bt2)[0]
    _close(l12.cpu(), (l1 + l2).cpu(), 1e-5, 'linearity in the top-down classifier')
    Xs = X[5:9].contiguous()      # (Xatt must be the same object as X for the fused path)
    ls = cof.attn_pool_fwd(Xs, Xs, Wa, ba, Wt, bt)[0]
    _close(ls.cpu(), l1[5:9].cpu(), 2e-6, 'per-image results must not depend on the batch')
    l1b = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt)[0]
    assert torch.equal(l1b, l1), 'the forward pass must be run-to-run deterministic'


def test_capi_error_paths(gpu):
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    X = torch.zeros(2, 4, 302, device=gpu)        # C = 302 is not a whole number of 16-byte vectors
    Wa = torch.zeros(302, 1, device=gpu); ba = torch.zeros(1, device=gpu)
    Wt = torch.zeros(302, 5, device=gpu); bt = torch.zeros(5, device=gpu)
    with pytest.raises(cof.ApaError, match='APA_ERR_UNSUPPORTED'):
        cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt)
    X4 = torch.zeros(2, 4, 300, device=gpu)       # C = 300 (a multiple of 4) runs on the generic kernels
    assert cof.attn_pool_fwd(X4, X4, Wa[:300].contiguous(), ba, Wt[:300].contiguous(), bt)[0].shape == (2, 5)
    with pytest.raises(cof.ApaError, match='GPU memory'):
        cof.attn_pool_fwd(X.cpu(), X.cpu(), Wa, ba, Wt, bt)


def test_device_side_dropout_counter_and_graph_replay(gpu):
    """APA_FLAG_RNG_DEVICE: the dropout step counter lives in HBM, the backward call advances it,
    so a captured hipGraph draws a fresh mask per replay and matches the by-value offsets."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    inp = make_head_inputs(N=4, H=7, W=7, C=2048, K=51, seed=13)
    X = inp['X'].to(gpu)
    Wa, ba, Wt, bt = (inp[k].to(gpu) for k in ('Wa', 'ba', 'Wt', 'bt'))
    labels = inp['labels'].to(gpu)
    flags = cof.attn_flags(False, False, True)

    def run(offset):
        logits, att, zs, ab, _, ws = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, flags=flags, keep_prob=0.5,
                                                       seed=7, offset=offset)
        _, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels)
        out = cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, att, zs, ab, G, flags=flags, keep_prob=0.5,
                                seed=7, offset=offset, workspace=ws)
        return logits, out[0]

    ref = [run(i) for i in range(3)]                      # offsets by value: 0, 1, 2
    ctr = torch.zeros(1, dtype=torch.int64, device=gpu)
    for i in range(3):                                    # same thing through the HBM counter
        lg, dX = run(ctr)
        assert torch.equal(lg, ref[i][0]) and torch.equal(dX, ref[i][1])
    torch.cuda.synchronize()
    assert int(ctr.item()) == 3
    assert not torch.equal(ref[0][0], ref[1][0])          # masks differ between steps

    ctr.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(ctr)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ctr.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lg_g, dX_g = run(ctr)
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(lg_g, ref[i][0]) and torch.equal(dX_g, ref[i][1]), 'replay {}'.format(i)


@pytest.mark.parametrize('N,K', [(1, 393), (33, 393), (70, 51), (40, 600), (130, 393), (257, 100), (300, 393),
                                 (520, 51)])
def test_m1_batch_and_class_tails(gpu, N, K):
    """Tile tails of the small MFMA kernels: N not a multiple of 32 (and > 64: multi-block
    cross-entropy), K not a multiple of 16/32/128, K > 512 (streaming softmax variant); large
    batches (many row tiles per block in the small MFMA kernels)."""
    inp = make_head_inputs(N=N, H=4, W=4, C=2048, K=K, seed=100 + N)
    ref = _oracle(inp, orc.AttnFlags())
    got = _run_hip(inp, gpu)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['loss'], ref['loss'], TIGHT, 'loss')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))
    _grads_close(got, ref, ('dX', 'dWa', 'dba', 'dWt', 'dbt'))


@pytest.mark.parametrize('target_blocks', [4, 12, 64])
@pytest.mark.parametrize('mode', ['id', 'softmax_dropout', 'sep_att', 'bf16', 'c1024'])
def test_m1_stream_kernels_pixel_chunking(gpu, monkeypatch, target_blocks, mode):
    """The channel-split streaming kernels process a block's pixels in chunks of <= 16.  Force
    S = 1 / 3 / 16 splits per image (13 / 5 / 1 chunks per block at 14x14) and check every variant
    against the oracle; APA_M1_STREAM=0 must give the same answer from the per-pixel kernels."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    monkeypatch.setenv('APA_M1_TARGET_BLOCKS', str(target_blocks))
    C = 1024 if mode == 'c1024' else 2048
    inp = make_head_inputs(N=4, H=14, W=14, C=C, K=51, Ca=768 if mode == 'sep_att' else None,
                           seed=77, dtype=torch.bfloat16 if mode == 'bf16' else torch.float32)
    softmax = mode in ('softmax_dropout', 'c1024')
    train = mode == 'softmax_dropout'
    keep, seed, offset = 0.5, 5, 11
    mask = cof.dropout_mask(tuple(inp['X'].shape), keep, seed, offset).cpu() if train else None
    flags = orc.AttnFlags(single_layer_att=(mode != 'sep_att'), softmax_att=softmax)
    ref = _oracle(inp, flags, train=train, keep=keep, mask=mask)
    got = _run_hip(inp, gpu, softmax=softmax, train=train, keep=keep, seed=seed, offset=offset)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['att'].reshape(ref['att'].shape), ref['att'], TIGHT, 'attention map')
    keys = ('dWa', 'dba', 'dWt', 'dbt') + (() if mode == 'bf16' else ('dX',))
    _grads_close(got, ref, keys + (('dXatt',) if mode == 'sep_att' else ()))
    if mode == 'bf16':
        err = (got['dX'].double() - ref['dX']).abs()
        assert float((err - ref['dX'].abs() * 2 ** -8).max()) <= 1e-5 * float(ref['dX'].abs().max())


def test_m1_full_baseline_workload_parity(gpu):
    """The bench workload itself (BASELINE configs[1]: N = 32 x 14x14x2048 fp32, K = 393, training
    dropout keep 0.2 -- the 512-block plan the timed kernels run) against the float64 oracle fed the
    kernel's own mask: logits within the north-star 1e-3, argmax bit-exact, every gradient tight."""
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    inp = make_head_inputs(N=32, H=14, W=14, C=2048, K=393, seed=42)
    keep, seed, offset = 0.2, 42, 5
    mask = cof.dropout_mask(tuple(inp['X'].shape), keep, seed, offset).cpu()
    ref = _oracle(inp, orc.AttnFlags(single_layer_att=True), train=True, keep=keep, mask=mask)
    got = _run_hip(inp, gpu, train=True, keep=keep, seed=seed, offset=offset)
    _close(got['logits'], ref['logits'], ATOL_LOGITS, 'logits (north_star tol)', absolute=True)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['loss'], ref['loss'], TIGHT, 'loss')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1)), 'argmax must be bit-exact'
    _grads_close(got, ref, ('dX', 'dWa', 'dba', 'dWt', 'dbt'))
    # run-to-run determinism of the whole step at this size
    got2 = _run_hip(inp, gpu, train=True, keep=keep, seed=seed, offset=offset)
    for k in ('logits', 'dX', 'dWa', 'dba', 'dWt', 'dbt'):
        assert torch.equal(got[k], got2[k]), k


@pytest.mark.parametrize('N,H,W,C,dt', [(2, 1, 1, 2048, torch.float32), (3, 2, 1, 2048, torch.float32),
                                        (1, 3, 3, 1024, torch.float32), (2, 5, 4, 1024, torch.bfloat16),
                                        (2, 3, 3, 512, torch.bfloat16), (1, 14, 14, 4096, torch.float32)])
def test_m1_edge_shapes(gpu, N, H, W, C, dt):
    """Degenerate maps (a single pixel, fewer pixels than a chunk), the narrow-map kernels in bf16,
    and the widest instantiated map (C = 4096)."""
    inp = make_head_inputs(N=N, H=H, W=W, C=C, K=51, seed=N * 100 + H * 10 + W, dtype=dt)
    softmax = H * W > 2      # (a softmax over one or two pixels has ~zero gradients: nothing to compare)
    ref = _oracle(inp, orc.AttnFlags(softmax_att=softmax))
    got = _run_hip(inp, gpu, softmax=softmax)
    _close(got['logits'], ref['logits'], TIGHT, 'logits')
    _close(got['att'].reshape(ref['att'].shape), ref['att'], TIGHT, 'attention map')
    assert torch.equal(got['pred'], ref['logits'].argmax(dim=1))
    keys = ('dWa', 'dba', 'dWt', 'dbt') + (() if dt == torch.bfloat16 else ('dX',))
    _grads_close(got, ref, keys)
