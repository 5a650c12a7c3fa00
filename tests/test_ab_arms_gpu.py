"""Launch-regrouping arms of the library must not change a single bit.

The product library has no run-time switches; the development build of the same sources (`make ABLATE=1`,
`libapa_hip_ablate.so`, built by `__graft_entry__.build()`) reads its A/B knobs from the environment.  Each arm below
only changes WHICH launch carries a piece of work -- two products in one launch (GemmDesc::twin), non-temporal instead of
plain stores, dropout(X) on the padding launch -- so the SHA-256 of every output of a per-class training step must be the
same with the arm on and off.  (Arms that change a summation order -- the folded activation passes -- are held to the
oracle instead: tests/test_bf16_parity_gpu.py, tools/fuzz_arms.sh.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, 'attentionalpoolingaction_amd', 'custom_ops', 'libapa_hip_ablate.so')


def _digest(shape, env):
    e = dict(os.environ, APA_LIB_PATH=ABL, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ab_digest.py')] + [str(a) for a in shape],
                       env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('DIGEST ')]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    return json.loads(lines[-1][7:])


ARMS = ({'APA_GEMM_TWIN': '0'}, {'APA_GEMM_NT': '0'}, {'APA_GEMM_TWIN': '0', 'APA_GEMM_NT': '0'})


@pytest.mark.parametrize('shape', [(3, 7, 512, 130), (4, 14, 2048, 393), (2, 5, 256, 70, 'relu')])
def test_generic_per_class_step_is_bit_identical_with_the_regrouping_arms_off(gpu, shape):
    if not os.path.exists(ABL):
        pytest.skip('development library not built (python -c "import __graft_entry__ as g; g.build()")')
    base = _digest(shape, {})
    for arm in ARMS:
        other = _digest(shape, arm)
        assert base == other, (arm, {k: (base[k], other[k]) for k in base if base[k] != other[k]})
