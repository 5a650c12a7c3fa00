"""numpy twin of the LIBRARY'S OWN dropout stream (csrc/apa_device.h: rng_key_dev, rng_hash, rng_keep2; the
threshold rule of csrc/apa_internal.h keep_thresh) -- test infrastructure.

Why it exists: the reference-generated head fixtures normally carry a mask drawn from a numpy stream, which the
product can only replay through APA_FLAG_RNG_EXTERNAL, i.e. through its generic kernels.  A fixture whose
recorded `tf.nn.dropout` uniforms are derived from THIS mask instead (make_head_reference.py, `libmask=`) lets
the product run its hot streaming kernels with their own counter hash -- seed and offset set to the fixture's
-- and still be compared with what the reference's code computed for that very mask.
`tests/test_reference_fixtures_gpu.py` checks the twin against `apa_dropout_mask` on the GPU bit for bit.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def rng_key(seed, offset):
    """splitmix64(seed, offset) -> (k0, k1); python integers, 64-bit wrap-around"""
    mask = (1 << 64) - 1
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(offset) + 1)) & mask
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    z ^= z >> 31
    return z & 0xFFFFFFFF, z >> 32


def _umul24(a, b):
    """v_mul_u32_u24: the low 24 bits of both operands, low 32 bits of the product"""
    return ((a & np.uint64(0xFFFFFF)) * (np.uint64(b) & np.uint64(0xFFFFFF))) & M32


def rng_hash(idx, k0, k1):
    x = (idx ^ np.uint64(k0)) & M32
    x ^= x >> np.uint64(16)
    x = _umul24(x, 0xB5297B)
    x ^= x >> np.uint64(13)
    x ^= k1
    x &= M32
    x ^= x >> np.uint64(17)
    x = _umul24(x, 0x68E31D)
    x ^= x >> np.uint64(15)
    return x


def keep_thresh(keep_prob):
    t = float(np.float32(keep_prob)) * 65536.0 + 0.5
    return int(min(max(t, 0.0), 65536.0))


def keep_mask(shape, keep_prob, seed, offset):
    """{0,1} uint8 mask of `shape` (flat element index = C-order position), as APA_FLAG_TRAIN applies it"""
    n = int(np.prod(shape))
    assert n % 2 == 0
    k0, k1 = rng_key(seed, offset)
    q = np.arange(n // 2, dtype=np.uint64)
    k1q = (np.uint64(k1) ^ _umul24(q >> np.uint64(32), 0x9E3779)) & M32
    h = rng_hash(q & M32, k0, k1q)
    th = np.uint64(keep_thresh(keep_prob))
    out = np.empty(n, dtype=np.uint8)
    out[0::2] = (h & np.uint64(0xFFFF)) < th
    out[1::2] = (h >> np.uint64(16)) < th
    return out.reshape(shape)
