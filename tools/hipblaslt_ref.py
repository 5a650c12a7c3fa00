"""The library yardstick for the dense rows: torch.matmul in bf16 on ROCm dispatches to hipBLASLt.  Times the
three products of the cfg 003 pose head (N = 32, 14x14: [6272 x 2048] x [2048 x 768] and its two backward forms)
and the K = 51 per-class shapes, to put this repository's hand-written MFMA kernels next to the vendor GEMM on the
same box.  Run through gpurun; the output is committed as profiles/<round>_hipblaslt_ref.log.

    python tools/hipblaslt_ref.py
"""
import torch, time
dev='cuda'
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t)/n*1e6
M,K,N=6272,2048,768
A=torch.randn(M,K,device=dev).bfloat16(); B=torch.randn(K,N,device=dev).bfloat16()
G=torch.randn(M,N,device=dev).bfloat16()
for name,fn,fl in [('fwd X.W1 [6272x2048]x[2048x768]', lambda: A@B, 2*M*K*N),
                ('dX dP.W1^T [6272x768]x[768x2048]', lambda: G@B.t(), 2*M*K*N),
                ('dW1 X^T.dP [2048x6272]x[6272x768]', lambda: A.t()@G, 2*M*K*N)]:
    us=bench(fn)
    print(name, '%.1f us'%us, '%.0f TFLOP/s'%(fl/us/1e6))
# the form the cfg 003 step actually runs: dX += dP.W1^T (beta = 1: the old 25.7 MB of dX are read as well)
Cacc=torch.randn(M,K,device=dev).bfloat16(); Bt=B.t().contiguous()      # [768, 2048]
us=bench(lambda: torch.addmm(Cacc, G, Bt, out=Cacc))
print('dX += dP.W1^T (addmm, beta=1) [6272x768]x[768x2048]', '%.1f us'%us, '%.0f TFLOP/s'%(2*M*K*N/us/1e6))
# per-class shapes K=51 padded 64
Wp=torch.randn(K,64,device=dev).bfloat16()
print('pc fwd [6272x2048]x[2048x64]', '%.1f us'%bench(lambda: A@Wp))
G2=torch.randn(M,64,device=dev).bfloat16()
print('pc dX [6272x64]x[64x2048]', '%.1f us'%bench(lambda: G2@Wp.t()))
print('pc dW [2048x6272]x[6272x64]', '%.1f us'%bench(lambda: A.t()@G2))
