"""Does the relative placement of X and dX change the streaming backward pass?  (N = 512: 1.64 GB each)"""
import sys, torch
sys.path.insert(0, '.')
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
dev = torch.device('cuda:0')
N, P, C, K = 512, 196, 2048, 393
g = torch.Generator().manual_seed(1)
nel = N * P * C
PAD = 64 << 20
xbuf = torch.empty(nel + PAD // 4, dtype=torch.float32, device=dev)
dbuf = torch.empty(nel + PAD // 4, dtype=torch.float32, device=dev)
print('base addresses', hex(xbuf.data_ptr()), hex(dbuf.data_ptr()), 'distance MB', (dbuf.data_ptr() - xbuf.data_ptr()) / 2 ** 20)
Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
labels = torch.randint(0, K, (N,), generator=g).to(dev)
flags = cof.attn_flags(False, False, True)
def run(xoff, doff, reps=6):
    X = xbuf[xoff // 4: xoff // 4 + nel].view(N, P, C)
    X.normal_().relu_()
    dX = dbuf[doff // 4: doff // 4 + nel].view(N, P, C)
    grads = (dX, None, torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt), torch.empty_like(bt))
    st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, flags=flags, keep_prob=0.2, seed=42, offset=0)
    kt = cof.KernelTimer(reps, None)
    for i in range(3): st.run()
    torch.cuda.synchronize()
    for i in range(reps): st.run(hooks=kt.hooks(i))
    torch.cuda.synchronize()
    f = sorted(kt.fwd_elapsed_ms())[reps // 2] * 1e3
    b = sorted(kt.bwd_elapsed_ms())[reps // 2] * 1e3
    return f, b
for rnd in range(2):
    for xoff, doff in ((0, 0), (0, 4096), (0, 65536), (0, 1 << 20), (0, (1 << 20) + 65536), (0, 33 << 20), (65536, 0), (1 << 20, 0), ((1 << 20) + 4096, 8192)):
        f, b = run(xoff, doff)
        print('xoff %9d doff %9d  fwd %.1f us  bwd %.1f us' % (xoff, doff, f, b), flush=True)
