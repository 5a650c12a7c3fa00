"""Random-shape sweeps (fixed seeds) of the C-ABI entry points against the CPU oracle: the shapes the
hand-written cases do not think of.  The generators live in tools/fuzz_attn_pool.py / tools/fuzz_all.py (run
them with more cases and other seeds when kernels change); two real bugs of a fallback kernel were found
this way in round 2 (tests/test_attn_pool_gpu.py::test_m1_backward_fallback_kernel_shapes_...)."""
import random

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('family,cases,seed', [('step', 25, 11), ('xent', 40, 11), ('perclass', 25, 11),
                                               ('pose', 25, 11), ('bf16m1', 25, 11), ('losses', 40, 11),
                                               ('wimg', 30, 11)])
def test_entry_points_on_random_shapes(gpu, family, cases, seed):
    from tools import fuzz_all
    rnd = random.Random(seed * 131 + len(family))
    for i in range(cases):
        try:
            fuzz_all.FAMILIES[family](rnd, i)
        except AssertionError as e:
            raise AssertionError('{} case {} {}: {}'.format(family, i, fuzz_all.LAST, e)) from e


def test_m1_pooling_on_random_shapes(gpu):
    import sys
    from tools import fuzz_attn_pool
    argv = sys.argv
    sys.argv = ['fuzz_attn_pool.py', '60', '23']
    try:
        assert fuzz_attn_pool.main() == 0
    finally:
        sys.argv = argv
