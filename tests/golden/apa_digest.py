"""Compact stand-ins for tensors too large to commit (test infrastructure, shared by the fixture generator
tests/golden/make_head_reference.py and the readers in tests/).

A tensor `t` of n elements is represented by
    digest(t)  = [sum t, sum t^2, <t, r1>, <t, r2>]      r_i = RandomState(0x5EED + i).standard_normal(n), float64
    sample(t)  = t.flat[sample_index(n)]                 4096 fixed positions
The two random projections see EVERY element: a wrong tile / block / row anywhere moves them by
|error|_2-sized amounts, while the sample pins individual values at full precision.
"""
import numpy as np

SAMPLE = 4096


def sample_index(n, k=SAMPLE):
    return np.sort(np.random.RandomState(4096).randint(0, n, size=k))


def sample(t):
    t = np.asarray(t).reshape(-1)
    return t[sample_index(t.size)].astype(np.float64)


def digest(t):
    t = np.asarray(t, dtype=np.float64).reshape(-1)
    out = [t.sum(), float(t @ t)]
    for i in (1, 2):
        out.append(float(t @ np.random.RandomState(0x5EED + i).standard_normal(t.size)))
    return np.array(out, dtype=np.float64)


def bf16_round(a):
    """round-to-nearest-even to bfloat16, returned as float64 values (finite inputs)"""
    u = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32).astype(np.uint64)
    u = (u + ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(a))


def check(got, dig, smp, tol_sample, tol_proj, what=''):
    """`got` (full tensor) against a stored digest + sample.  sample: max abs error <= tol_sample * max|sample|;
    projections and sum: |delta| <= tol_proj * sqrt(sum t^2) * (1 resp. sqrt(n) for the plain sum);
    sum of squares: relative 2 * tol_proj."""
    g = np.asarray(got, dtype=np.float64).reshape(-1)
    s = g[sample_index(g.size)]
    scale = max(float(np.abs(smp).max()), 1e-30)
    err = float(np.abs(s - smp).max())
    assert err <= tol_sample * scale, '%s sample: %.3e > %.1e * %.3e' % (what, err, tol_sample, scale)
    d = digest(g)
    l2 = float(np.sqrt(dig[1]))
    assert abs(d[0] - dig[0]) <= tol_proj * l2 * np.sqrt(g.size), '%s sum: %r vs %r' % (what, d[0], dig[0])
    assert abs(d[1] - dig[1]) <= 2 * tol_proj * dig[1], '%s sum of squares: %r vs %r' % (what, d[1], dig[1])
    for i in (2, 3):
        assert abs(d[i] - dig[i]) <= tol_proj * l2, '%s projection %d: %r vs %r (l2 %r)' % (what, i, d[i], dig[i], l2)
