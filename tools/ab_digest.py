#!/usr/bin/env python
"""SHA-256 digests of every output of one per-class attention-head training step (one-call entry point), as one JSON
line.  Run in a subprocess per A/B arm of the development library (APA_LIB_PATH = libapa_hip_ablate.so, knobs in the
environment) by tests/test_ab_arms_gpu.py: arms that only regroup launches must give identical digests.

    python tools/ab_digest.py N H C K [relu]"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    N, H, C, K = (int(a) for a in sys.argv[1:5])
    relu = len(sys.argv) > 5 and sys.argv[5] == 'relu'
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(100 * K + N)
    X = torch.relu(torch.randn(N, H * H, C, generator=g)).bfloat16().to(dev)
    Wa = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev)
    ba = (torch.randn(K, generator=g) * 0.1).to(dev)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev)
    bt = (torch.randn(K, generator=g) * 0.1).to(dev)
    labels = torch.randint(0, K, (N,), generator=g).to(dev)
    grads = (torch.empty_like(X), None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt),
             torch.empty_like(bt))
    st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, flags=cof.attn_flags(False, relu, True),
                           keep_prob=0.5, seed=9, offset=4)
    st.run()
    torch.cuda.synchronize()
    out = {'logits': st.logits, 'att': st.att, 'loss': st.loss, 'G': st.G, 'dX': grads[0], 'dWa': grads[2],
           'dba': grads[3], 'dWt': grads[4], 'dbt': grads[5]}
    dig = {}
    for k, v in out.items():
        t = v.detach().contiguous().cpu()
        t = t.view(torch.int16) if t.dtype == torch.bfloat16 else t
        dig[k] = hashlib.sha256(t.numpy().tobytes()).hexdigest()[:16]
    print('DIGEST ' + json.dumps(dig, sort_keys=True))


if __name__ == '__main__':
    main()
