import sys, torch
sys.path.insert(0, '.')
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for N, C, K in ((32, 2048, 393), (5, 2048, 51), (3, 512, 12)):
    z = torch.randn(N, 1, C, generator=g).to(dev)
    ones_in = torch.zeros(N, 1, 8, device=dev)
    Wa = torch.zeros(8, 1, device=dev); ba = torch.ones(1, device=dev)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.randn(K, generator=g).to(dev)
    for train in (False, True):
        flags = cof.attn_flags(False, False, train)
        logits, att, zs, ab, _, ws = cof.attn_pool_fwd(z, ones_in, Wa, ba, Wt, bt, flags=flags, keep_prob=0.5, seed=3, offset=1)
        mask = cof.dropout_mask(N * C, 0.5, 3, 1).view(N, C).float() if train else torch.ones(N, C, device=dev)
        want = (z.view(N, C) * mask / (0.5 if train else 1.0)).double() @ Wt.double() + bt.double()
        print(N, C, K, train, float((logits.double() - want).abs().max()))
        G = torch.randn(N, K, generator=g).to(dev)
        dX, dXatt, dWa, dba, dWt, dbt = cof.attn_pool_bwd(z, ones_in, Wa, ba, Wt, bt, att, zs, ab, G, flags=flags, keep_prob=0.5, seed=3, offset=1, workspace=ws)
        wdz = (G.double() @ Wt.double().t()) * mask / (0.5 if train else 1.0)
        print('   dz', float((dX.view(N, C).double() - wdz).abs().max()), 'dWt', float((dWt.double() - (z.view(N, C) * mask / (0.5 if train else 1.0)).double().t() @ G.double()).abs().max()))
