"""The one-block-per-CU ring GEMM (csrc/apa_gemm_bf16.hip: gemm_bf16_ring_kernel) through the pose-head entry point:
every tile height MT = 4 .. 8 (3- and 4-stage rings), odd and minimal K-tile counts, a ragged last row tile and a
ragged last column tile, against a float64 product of the same bf16-rounded operands.  The shapes are chosen so
that `ring_pick_mt` (tiles = ceil(M / 32 MT) * ceil(N / 128) <= 256 CUs) lands on the MT named in the id."""
import numpy as np
import pytest
import torch

from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof

pytestmark = pytest.mark.gpu

#          id                 N   P    C     Cp
SHAPES = [('mt4_nk5',         32, 196, 320,  448),
          ('mt5_nk4',         32, 196, 256,  768),
          ('mt5_nk32',        32, 196, 2048, 768),
          ('mt6_nk7',         32, 196, 448,  896),
          ('mt7_nk5',         32, 196, 320,  1024),
          ('mt7_nk6_ragged',  31, 193, 384,  1100),      # M = 5983: ragged last row tile; 1100: ragged column tile
          ('mt8_nk5',         32, 196, 320,  1280),
          ('mt4_small',       5,  49,  512,  200)]


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize('case', SHAPES, ids=[s[0] for s in SHAPES])
def test_ring_gemm_tile_heights_through_the_pose_head(gpu, case):
    _, N, P, C, Cp = case
    J = 16
    g = torch.Generator().manual_seed(C * 7 + Cp)
    X = (torch.randn(N, P, C, generator=g) * 0.7).to(torch.bfloat16)
    W1 = torch.randn(C, Cp, generator=g) / np.sqrt(C)
    b1 = torch.randn(Cp, generator=g) * 0.1
    W2 = torch.randn(Cp, J, generator=g) / np.sqrt(Cp)
    b2 = torch.randn(J, generator=g) * 0.1
    Ppre, Pl, _ = cof.pose_head_fwd(X.to(gpu), W1.to(gpu), b1.to(gpu), W2.to(gpu), b2.to(gpu))
    torch.cuda.synchronize()
    ref = torch.relu(X.to(torch.float64).reshape(-1, C) @ _bf16_round(W1) + b1.double())
    got = Ppre.float().cpu().double().reshape(-1, Cp)
    # fp32 accumulation of exact bf16 products, one bf16 rounding of the result (2^-9 relative, half an ulp)
    err = (got - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 2e-4
    assert bool((err <= bound).all()), float((err - bound).max())
    assert float(got.abs().max()) > 1.0                      # the comparison is not vacuous
    # the second layer consumes the stored (bf16) pre-logits
    ref_pl = got @ W2.double() + b2.double()
    assert float((Pl.cpu().double().reshape(-1, J) - ref_pl).abs().max()) <= 3e-3 * float(ref_pl.abs().max())


#               id                  N   P    C     Cp    (M = N P rows, C output columns >= 1024, Cp / 64 K tiles <= 16)
WIDE_SHAPES = [('wide_mt7_nk12',    32, 196, 2048, 768),      # the cfg 003 dX product: 28 x 8 = 224 tiles
               ('wide_mt4_nk12',    16, 196, 2048, 768),      # 25 x 8 = 200 tiles of 128 x 256
               ('wide_mt5_nk4',     26, 196, 2048, 256),      # M = 5096: 32 x 8 tiles of 160 x 256
               ('wide_mt6_ragged',  31, 193, 2048, 640),      # M = 5983: ragged last row tile (192-row tiles)
               ('wide_mt4_ragged_n', 32, 196, 1152, 128)]     # 4.5 column tiles, two K tiles (the shortest loop)


@pytest.mark.parametrize('case', WIDE_SHAPES, ids=[s[0] for s in WIDE_SHAPES])
def test_wide_gemm_through_the_pose_head_backward(gpu, case):
    """gemm_bf16_wide_kernel ((32 MT) x 256 tiles, one resident round, two LDS stages) as the pose head's
    dX (+)= dPpre . W1^T: every tile height, a ragged row tile, a ragged column tile, overwrite and accumulate,
    against a float64 product of the operands the kernel reads (dPpre as stored in bf16 is re-derived from
    dPl . W2^T masked by Ppre > 0; its own rounding is inside the bound)."""
    _, N, P, C, Cp = case
    J = 16
    g = torch.Generator().manual_seed(C * 3 + Cp + N)
    X = (torch.randn(N, P, C, generator=g) * 0.7).to(torch.bfloat16)
    W1 = torch.randn(C, Cp, generator=g) / np.sqrt(C)
    b1 = torch.randn(Cp, generator=g) * 0.1
    W2 = torch.randn(Cp, J, generator=g) / np.sqrt(Cp)
    b2 = torch.randn(J, generator=g) * 0.1
    dPl = torch.randn(N, P, J, generator=g)
    Xd, W1d, W2d = X.to(gpu), W1.to(gpu), W2.to(gpu)
    Ppre, Pl, ws = cof.pose_head_fwd(Xd, W1d, b1.to(gpu), W2d, b2.to(gpu))
    dX, dW1, db1, dW2, db2 = cof.pose_head_bwd(Xd, W1d, W2d, Ppre, dPl.to(gpu), None, workspace=ws)
    torch.cuda.synchronize()
    pre = Ppre.float().cpu().double().reshape(-1, Cp)
    dpre = (dPl.double().reshape(-1, J) @ W2.double().t()) * (pre > 0)
    w1b = _bf16_round(W1)                                              # [C, Cp]
    ref = _bf16_round(dpre.float()) @ w1b.t()                          # [M, C]
    got = dX.float().cpu().double().reshape(-1, C)
    # dPpre is rounded to bf16 when stored (2^-9 relative each), the result once more
    bound = 2.0 ** -8 * (dpre.abs() @ w1b.abs().t()) + 2.0 ** -8 * ref.abs() + 1e-4
    err = (got - ref).abs()
    assert bool((err <= bound).all()), (float((err - bound).max()), int((err > bound).sum()))
    assert float(got.abs().max()) > 0.1
    # accumulate form (beta = 1): dX2 = base + product; and dW1 = X^T dPpre while we are here
    base = torch.randn(N, P, C, generator=g).to(torch.bfloat16)
    dX2, *_ = cof.pose_head_bwd(Xd, W1d, W2d, Ppre, dPl.to(gpu), None, dX=base.to(gpu).clone(), accumulate_dX=True)
    want2 = base.double().reshape(-1, C) + ref
    err2 = (dX2.float().cpu().double().reshape(-1, C) - want2).abs()
    assert bool((err2 <= bound + 2.0 ** -8 * want2.abs()).all()), float((err2 - bound).max())
    ref_dw1 = X.double().reshape(-1, C).t() @ _bf16_round(dpre.float())
    assert float((dW1.cpu().double() - ref_dw1).abs().max()) <= 2e-2 * float(ref_dw1.abs().max())
