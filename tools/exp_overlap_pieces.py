"""Experiment behind DESIGN.md section 5: what each piece of the two-stream gradient-sum schedule costs on ONE
rank (collectives are no-ops there): library hooks, cross-stream event waits, RCCL calls.  Run through
gpurun:  python tools/exp_overlap_pieces.py"""
import sys, time, torch
sys.path.insert(0, '.')
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
from attentionalpoolingaction_amd import rccl
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
c1, c2 = rccl.RcclCommunicator(0, 1, dev), rccl.RcclCommunicator(0, 1, dev)
side = torch.cuda.Stream()
ready, done = torch.cuda.Event(), torch.cuda.Event()
ready.record(); done.record(); torch.cuda.synchronize()
main = torch.cuda.current_stream()

def bench(name, after, steps=200):
    for _ in range(20):
        st.run(); after()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.run(); after()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # short run for host time
    t3 = time.perf_counter()
    for _ in range(30):
        st.run(); after()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    print('%-46s wall %6.1f us/step   host %5.1f us/step' % (name, (t2 - t0) / steps * 1e6, (t4 - t3) / 30 * 1e6))

bench('plain', lambda: None)
st.hooks = cof.make_hooks(grad_ready=ready)
bench('+ready record', lambda: None)
def a1():
    side.wait_event(ready)
bench('+side.wait_event(ready)', a1)
def a2():
    side.wait_event(ready); done.record(side)
bench('+done.record(side)', a2)
st.hooks = cof.make_hooks(grad_ready=ready, td_weights_ready=done)
bench('+fwd waits done', a2)
def a3():
    side.wait_event(ready); c2.all_reduce_(b_td, side); done.record(side)
bench('+AR td on side', a3)
def a4():
    side.wait_event(ready); c2.all_reduce_(b_td, side); done.record(side); c1.all_reduce_(b_att, main)
bench('+AR att on main (full schedule)', a4)
st.hooks = None
bench('plain again', lambda: None)
def a5():
    c1.all_reduce_(bucket, main)
bench('one in-stream AR', a5)
