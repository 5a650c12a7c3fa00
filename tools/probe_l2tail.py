"""Upper bound of what the forward pass's tail in L2 is worth to the backward kernel (round 6; docs/DESIGN_HISTORY.md)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
dev = torch.device('cuda:0')
N, P, C, K = 32, 196, 2048, 393
R = 8
Wa = (torch.randn(C, 1) / C ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
Wt = (torch.randn(C, K) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
labels = torch.randint(0, K, (N,)).to(dev)
flags = cof.attn_flags(False, False, True)
bucket = torch.zeros(C + 1 + C * K + K, device=dev)
dWa, dba, dWt, dbt = bucket[:C].view(C, 1), bucket[C:C + 1], bucket[C + 1:C + 1 + C * K].view(C, K), bucket[C + 1 + C * K:]
ws = torch.empty((cof.attn_pool_workspace_bytes(N, P, C, C, K, 1, flags),), dtype=torch.uint8, device=dev)
ctr = torch.zeros(1, dtype=torch.int64, device=dev)
X = [torch.relu(torch.randn(N, P, C, device=dev)) for _ in range(R)]
dX = [torch.empty_like(x) for x in X]

def run(mode, n=40):
    timer = cof.KernelTimer(n)
    for i in range(n + 5):
        r = i % R
        h = timer.hooks(i - 5) if i >= 5 else None
        logits, att, zsave, abar, _, _ = cof.attn_pool_fwd(X[r], X[r], Wa, ba, Wt, bt, flags=flags, keep_prob=0.5, seed=42, offset=ctr, workspace=ws)
        _, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels, grad_scale=1.0)
        if mode == 'refwd':
            cof.attn_pool_fwd(X[r], X[r], Wa, ba, Wt, bt, flags=flags, keep_prob=0.5, seed=42, offset=ctr, workspace=ws)
        elif mode == 'other':   # a forward pass over ANOTHER map: same work in between, tail of the wrong map in L2
            cof.attn_pool_fwd(X[(r + 3) % R], X[(r + 3) % R], Wa, ba, Wt, bt, flags=flags, keep_prob=0.5, seed=42, offset=ctr, workspace=ws)
            # (att/zsave now belong to the other map: timing only)
        cof.attn_pool_bwd(X[r], X[r], Wa, ba, Wt, bt, att, zsave, abar, G, flags=flags, keep_prob=0.5, seed=42, offset=ctr,
                          workspace=ws, out=(dX[r], None, dWa, dba, dWt, dbt), hooks=h)
    torch.cuda.synchronize()
    b = timer.bwd_elapsed_ms()
    timer.close()
    return sorted(b)[len(b) // 2] * 1e3
for rep in range(3):
    print('lib', os.environ.get('APA_LIB_PATH', 'default'), ' plain %.2f  refwd %.2f  other-map fwd %.2f' % (run('plain'), run('refwd'), run('other')), flush=True)
