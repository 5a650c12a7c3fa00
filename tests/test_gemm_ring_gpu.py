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
