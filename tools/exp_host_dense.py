"""Host time to enqueue one step of a tools/bench_dense.py workload vs its wall time per step."""
import sys, time, torch
sys.path.insert(0, '.')
from tools import bench_dense as bd
from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
dev = torch.device('cuda:0')
for name, builder in (('cfg003', bd.build_cfg003), ('perclass', bd.build_perclass)):
    step, info = builder(cof, dev)
    for _ in range(30): step()
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / 20 * 1e6, (t2 - t0) / 20 * 1e6))
    res.sort()
    print(name, 'host enqueue %.1f us/step, wall %.1f us/step' % res[2])
