import sys, torch
sys.path.insert(0, '.')
from tests._synth import make_head_inputs
from tests.test_attn_pool_gpu import _oracle, _run_hip
from oracle import attn_pool_oracle as orc
gpu = torch.device('cuda:0')
inp = make_head_inputs(N=4, H=14, W=14, C=2048, K=393, seed=5, dtype=torch.bfloat16)
ref = _oracle(inp, orc.AttnFlags())
got = _run_hip(inp, gpu)
d = (got['dX'].double() - ref['dX'])
err = d.abs()
i = err.argmax()
idx = torch.unravel_index(i, err.shape)
print('max err', err.max().item(), 'at', [int(t) for t in idx], 'got', got['dX'].double().flatten()[i].item(), 'ref', ref['dX'].flatten()[i].item())
print('ref absmax', ref['dX'].abs().max().item(), 'got absmax', got['dX'].double().abs().max().item())
# per-channel-group error
e = err.reshape(-1, 2048).max(0).values
print('chan err first 32', e[:32])
print('chan err by lane-slot (c%8):', [float(e[k::8].max()) for k in range(8)])
print('chan err by 512-block:', [float(e[k*512:(k+1)*512].max()) for k in range(4)])
rel = (err / ref['dX'].abs().clamp_min(1e-12))
print('frac of elements with rel err > 1%:', float((rel > 0.01).double().mean()))
