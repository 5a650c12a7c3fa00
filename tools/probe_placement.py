"""Where X and dX sit, and what that does to the backward streaming kernel (round 6; docs/DESIGN_HISTORY.md).

    python tools/probe_placement.py [--batch 512] [--trials 6] [--dtype f32|bf16]       (on the GPU box)

Per trial: one (X, dX) pair from the default allocator and one physically contiguous pair
(hipExtMallocWithFlags(hipDeviceMallocContiguous)), every pair freed again, a junk allocation of growing size in
between so that the driver hands out different physical pages.  Prints the two streaming kernels' durations (HIP event
pairs of the library, median of 8 steps).  Contiguous pairs reproduce to +-1 %; default pairs show the placement
lottery (N = 512 fp32: 275 ... 328 us for one binary in one process).  A/B of two builds: run it once per library
with APA_LIB_PATH=... (the walk direction of the backward pass: build apa_m1_stream.hip with -DAPA_M1S_BWD_DOWN=0).
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof  # noqa: E402

HIP_DEVICE_MALLOC_CONTIGUOUS = 0x4


class RawBuffer:
    """A hipExtMallocWithFlags allocation seen by torch through __cuda_array_interface__."""

    def __init__(self, hip, nbytes, flags):
        self.hip, self.p = hip, ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(self.p), nbytes, flags)
        if rc != 0:
            raise RuntimeError('hipExtMallocWithFlags({}, {}) -> {}'.format(nbytes, flags, rc))
        self.__cuda_array_interface__ = {'shape': (nbytes // 4,), 'typestr': '<f4', 'data': (self.p.value, False),
                                         'version': 2}

    def free(self):
        self.hip.hipFree(self.p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--hw', type=int, default=14)
    ap.add_argument('--trials', type=int, default=6)
    ap.add_argument('--dtype', choices=('f32', 'bf16'), default='f32')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    N, P, C, K = args.batch, args.hw * args.hw, 2048, 393
    dt = torch.float32 if args.dtype == 'f32' else torch.bfloat16
    esz = 4 if args.dtype == 'f32' else 2
    nel = N * P * C
    Wa = (torch.randn(C, 1) / C ** 0.5).to(dev)
    ba = torch.zeros(1, device=dev)
    Wt = (torch.randn(C, K) / C ** 0.5).to(dev)
    bt = torch.zeros(K, device=dev)
    labels = torch.randint(0, K, (N,)).to(dev)
    flags = cof.attn_flags(False, False, True)
    bucket = torch.zeros(C + 1 + C * K + K, device=dev)
    o = C + 1
    grads = (bucket[:C].view(C, 1), bucket[C:o], bucket[o:o + C * K].view(C, K), bucket[o + C * K:])
    ws = torch.empty((cof.attn_pool_workspace_bytes(N, P, C, C, K, 1, flags),), dtype=torch.uint8, device=dev)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)

    def measure(x, dx, n=8):
        st = cof.HeadTrainStep(x, x, Wa, ba, Wt, bt, labels, (dx, None) + grads, flags=flags, keep_prob=0.5, seed=42,
                               offset=ctr, grad_scale=1.0, workspace=ws)
        for _ in range(3):
            st.run()
        torch.cuda.synchronize()
        timer = cof.KernelTimer(n)
        for i in range(n):
            st.run(hooks=timer.hooks(i))
        torch.cuda.synchronize()
        f, b = sorted(timer.fwd_elapsed_ms()), sorted(timer.bwd_elapsed_ms())
        timer.close()
        return f[n // 2] * 1e3, b[n // 2] * 1e3

    print('library:', os.environ.get('APA_LIB_PATH', '(default)'), ' N', N, 'P', P, args.dtype)
    junk = []
    for t in range(args.trials):
        for name, fl in (('default', 0), ('contiguous', HIP_DEVICE_MALLOC_CONTIGUOUS)):
            a, b = RawBuffer(hip, nel * esz, fl), RawBuffer(hip, nel * esz, fl)
            x = torch.as_tensor(a, device=dev).view(dt).view(N, P, C)
            dx = torch.as_tensor(b, device=dev).view(dt).view(N, P, C)
            x.copy_(torch.relu(torch.randn(N, P, C, device=dev)))
            f, bw = measure(x, dx)
            print('trial %d  %-10s  forward %.1f us  backward %.1f us' % (t, name, f, bw), flush=True)
            del x, dx
            torch.cuda.synchronize()
            a.free()
            b.free()
        junk.append(RawBuffer(hip, (5 + 11 * t) << 20, 0))
    for j in junk:
        j.free()


if __name__ == '__main__':
    main()
