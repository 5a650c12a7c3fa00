import sys, time, ctypes, torch
sys.path.insert(0, '.')
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
dev = torch.device('cuda', 0)
N, P, C, K = 32, 196, 2048, 393
g = torch.Generator().manual_seed(1)
X = torch.relu(torch.randn(N, P, C, generator=g)).to(dev)
Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).to(dev); ba = torch.zeros(1, device=dev)
Wt = (torch.randn(C, K, generator=g) / C ** 0.5).to(dev); bt = torch.zeros(K, device=dev)
labels = torch.randint(0, K, (N,), generator=g).to(dev)
bucket = torch.zeros(C + 1 + C * K + K, device=dev)
b_att, b_td = bucket[:C + 1], bucket[C + 1:]
grads = (torch.empty_like(X), None, b_att[:C].view(C, 1), b_att[C:], b_td[:C * K].view(C, K), b_td[C * K:])
ctr = torch.zeros(1, dtype=torch.int64, device=dev)
flags = cof.attn_flags(False, False, True)
st = cof.HeadTrainStep(X, X, Wa, ba, Wt, bt, labels, grads, flags=flags, keep_prob=0.2, seed=42, offset=ctr)
hip = ctypes.CDLL([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0])
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
vp = ctypes.c_void_p

def mk(fl):
    e = vp()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(fl)) == 0
    assert hip.hipEventRecord(e, vp(main.cuda_stream)) == 0
    return e

def bench(name, after, steps=300):
    for _ in range(20):
        st.run(); after()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.run(); after()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-50s wall %6.1f us/step' % (name, (t2 - t0) / steps * 1e6))

bench('plain', lambda: None)
for fl, nm in ((0x2, 'DisableTiming'), (0x2 | 0x40000000, 'DisableTiming|ReleaseToDevice'),
               (0x2 | 0x20000000, 'DisableTiming|DisableSystemFence'), (0x2 | 0x60000000, 'all three'), (0x0, 'default(timing)')):
    ready, done = mk(fl), mk(fl)
    torch.cuda.synchronize()
    cof.set_grad_ready_event(ready.value)
    bench(nm + ': ready record only', lambda: None)
    cof.set_td_weights_ready_event(done.value)
    def a():
        hip.hipStreamWaitEvent(vp(side.cuda_stream), ready, 0)
        hip.hipEventRecord(done, vp(side.cuda_stream))
    bench(nm + ': full choreography', a)
    cof.set_grad_ready_event(None); cof.set_td_weights_ready_event(None)
    torch.cuda.synchronize()
bench('plain', lambda: None)
