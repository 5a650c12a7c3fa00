"""Random-shape parity sweep of the M == 1 pooling op against the CPU oracle (a tool, not a test: run it
through gpurun when kernels change).  python tools/fuzz_attn_pool.py [cases] [seed]"""
import random
import sys
import time
import traceback

import torch

sys.path.insert(0, '.')
from oracle import attn_pool_oracle as orc                      # noqa: E402
from tests import test_attn_pool_gpu as T                       # noqa: E402
from tests._synth import make_head_inputs                       # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    gpu = torch.device('cuda:0')
    bad = 0
    t0 = time.time()
    for i in range(cases):
        C = rnd.choice([8, 16, 64, 96, 256, 512, 832, 1000, 1024, 1280, 2048, 2064, 4096])
        K = rnd.choice([1, 2, 7, 16, 33, 51, 64, 101, 393, 400, 513, 600, 833, 1000, 2049, 5000])
        N = rnd.choice([1, 2, 3, 5, 8, 17, 32, 40, 65, 130, 600])
        H, W = rnd.choice([1, 2, 3, 7, 14, 15, 20]), rnd.choice([1, 2, 7, 14, 15])
        while N * H * W * C * K > 6e9:
            N = max(1, N // 2)
            if N == 1:
                K = max(1, K // 2)
        Ca = rnd.choice([None, None, 8, 200, 768])
        softmax, relu = rnd.choice([(False, False), (True, False), (False, True), (True, True)])
        train = rnd.random() < 0.5
        keep = rnd.choice([0.2, 0.5, 0.9])
        desc = dict(N=N, H=H, W=W, C=C, K=K, Ca=Ca, softmax=softmax, relu=relu, train=train, keep=keep)
        try:
            inp = make_head_inputs(N=N, H=H, W=W, C=C, K=K, Ca=Ca, seed=1000 + i)
            seed, offset = 7 + i, i % 5
            mask = cof.dropout_mask(tuple(inp['X'].shape), keep, seed, offset).cpu() if train else None
            fused = inp['Xatt'] is inp['X']
            flags = orc.AttnFlags(single_layer_att=fused, softmax_att=softmax, relu_att=relu)
            ref = T._oracle(inp, flags, train=train, keep=keep, mask=mask)
            got = T._run_hip(inp, gpu, softmax=softmax, relu=relu, train=train, keep=keep, seed=seed, offset=offset)
            T._close(got['logits'], ref['logits'], T.TIGHT, 'logits')
            T._close(got['att'].reshape(ref['att'].shape), ref['att'], T.TIGHT, 'attention map')
            T._close(got['loss'], ref['loss'], T.TIGHT, 'loss')
            # gradients: relative to the largest gradient of the same kind (softmax makes d(ba) exactly 0, and
            # tiny K makes dX tiny: a floor of 1e-6 x the scale of dWt keeps fp32 noise out of the verdict)
            floor = 1e-6 * float(ref['dWt'].abs().max())
            for k in ('dX', 'dWa', 'dba', 'dWt', 'dbt') + (() if fused else ('dXatt',)):
                # (d(ba) is ONE number, the sum of all dZ: with the softmax it cancels exactly, without it it may cancel
                #  to a small value -- N=3, 3x14, C=1000, K=5000, seed 41: |d(ba)| = 0.004 from 126 terms -- so its noise
                #  scales with |dZ|, i.e. with dWa, not with its own size)
                at = max(floor, 1e-5 * float(ref['dWa'].abs().max())) if k == 'dba' else floor
                T._close(got[k].reshape(ref[k].shape), ref[k], 5e-5, k, atol=at)
        except Exception as e:                                   # noqa: BLE001
            bad += 1
            print('FAIL', desc, type(e).__name__, str(e)[:300])
            if not isinstance(e, AssertionError):
                traceback.print_exc(limit=2)
    print('{} cases, {} failed, {:.0f} s'.format(cases, bad, time.time() - t0))
    return bad


if __name__ == '__main__':
    main()
