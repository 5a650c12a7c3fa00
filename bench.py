#!/usr/bin/env python
"""bench.py -- attentional-pooling head, forward + loss + backward, on synthetic conv5 features.

Metric (BASELINE.json): images/sec attn-pool fwd+bwd, N x 14 x 14 x 2048, 393 classes.
One "step" = one pass of the hot path over one batch that is already resident in HBM:
    apa_attn_pool_fwd  ->  apa_softmax_xent_fwd_bwd  ->  apa_attn_pool_bwd  [-> RCCL all-reduce of
    the flat head-gradient bucket when world_size > 1]
Workload (cfg 002 semantics): per-GPU batch 32, 14x14x2048 fp32 features, K=393,
class-agnostic bottom-up map (M=1), training-mode dropout keep=0.2 (nets_factory.py:145).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (m1_bwd_main_kernel: reads X
once, writes dX once); `cpu_baseline` is the literal op-by-op PyTorch-CPU restatement of the
reference graph (oracle/, a *port* -- TF1 cannot run here) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (cfg 002: 32)')
    ap.add_argument('--hw', type=int, default=14)
    ap.add_argument('--channels', type=int, default=2048)
    ap.add_argument('--classes', type=int, default=393)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--keep-prob', type=float, default=0.2)
    ap.add_argument('--eval-mode', action='store_true', help='no dropout (is_training=False)')
    ap.add_argument('--softmax-att', action='store_true')
    ap.add_argument('--graph', action='store_true',
                    help='capture the step in a hipGraph (measured: replay overhead makes it ~5%% '
                         'slower than eager launches for this 8-kernel step, so off by default)')
    ap.add_argument('--overlap', default='auto', choices=['auto', 'on', 'off'],
                    help='N > 1: reduce dWt|dbt (99.7 %% of the bytes) on a communication stream with its own '
                         'RCCL communicator, between the library\'s grad-ready and td-weights-ready hooks '
                         '(hidden under the streaming backward pass and the next pooling pass), and only '
                         'dWa|dba (8 KB) on the compute stream (deploy.OverlappedGradientSum).  auto = on '
                         'with --comm rccl, off (one in-stream bucket) with --comm torch, whose extra host '
                         'calls cost more than the overlap buys at this step size')
    ap.add_argument('--comm', default='rccl', choices=['rccl', 'torch', 'gloo'],
                    help='N > 1 gradient sum: direct in-stream ncclAllReduce through librccl (default), '
                         'torch.distributed.all_reduce over RCCL, or over gloo (host-staged; lets two ranks '
                         'share ONE GPU so the whole N > 1 flow can be exercised on a single-GPU box -- a '
                         'functional check, not a performance path)')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the N > 1 code path (RCCL group, side stream, split all-reduce) on one GPU')
    ap.add_argument('--relu-input', action='store_true',
                    help='APA_FLAG_RELU_INPUT: feed the pre-activation map (randn, not rectified) and let the '
                         'op apply the backbone\'s last ReLU on the fly in both passes')
    ap.add_argument('--per-op-calls', action='store_true',
                    help='drive the step as three separately marshalled calls (apa_attn_pool_fwd, '
                         'apa_softmax_xent_fwd_bwd, apa_attn_pool_bwd) with per-step output allocation instead '
                         'of one apa_attn_head_train_step call: same kernels, more host time per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run')
    return ap.parse_args()


def cpu_baseline(args, seconds):
    """Literal reference formulation (materialises T = X' . Wt like the TF graph) on host cores."""
    from oracle import attn_pool_oracle as orc
    g = torch.Generator().manual_seed(42)
    N, H, C, K = args.batch, args.hw, args.channels, args.classes
    X = torch.relu(torch.randn(N, H, H, C, generator=g)).requires_grad_(True)
    Wa = (torch.randn(C, 1, generator=g) / C ** 0.5).requires_grad_(True)
    ba = torch.zeros(1, requires_grad=True)
    Wt = (torch.randn(C, K, generator=g) / C ** 0.5).requires_grad_(True)
    bt = torch.zeros(K, requires_grad=True)
    labels = torch.randint(0, K, (N,), generator=g)
    train = not args.eval_mode
    mask = (torch.rand(N, H, H, C, generator=g) < args.keep_prob) if train else None
    flags = orc.AttnFlags(softmax_att=args.softmax_att)

    def it():
        for t in (X, Wa, ba, Wt, bt):
            t.grad = None
        logits, _ = orc.attentional_pooling(X, None, None, [Wa], [ba], [Wt], [bt], flags,
                                            is_training=train, keep_prob=args.keep_prob,
                                            dropout_mask=mask)
        orc.action_softmax_xent(logits, labels, K).backward()

    # 256 MKL threads on a 256-core box thrash on this problem size: probe a few thread counts
    # and keep the fastest (the count actually used is what `cores` reports)
    best_t, best_thr = None, None
    for thr in sorted({min(os.cpu_count(), c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(thr)
        it()
        t0 = time.perf_counter()
        it()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_thr = dt, thr
    torch.set_num_threads(best_thr)
    n, t0 = 0, time.perf_counter()
    while True:
        it()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds and n >= 3:
            break
    return {'value': round(N * n / el, 2), 'unit': 'images/sec', 'cores': best_thr,
            'kind': 'port',
            'sample': '{} iterations ({:.1f} s) of the same workload (N={}, {}x{}x{}, K={}, fp32, '
                      'literal [N,P,K] top-down formulation, torch {} CPU, best of 8..128 threads = {} on a '
                      '{}-core host)'.format(n, el, N, H, H, C, K, torch.__version__, best_thr,
                                             os.cpu_count())}


def _claim_stdout():
    """stdout must carry exactly one JSON line, but gloo ('[Gloo] Rank 0 is connected ...') and RCCL
    ('RCCL version : ...', 'Librccl path : ...') print banners through C stdio, some of them only
    flushed at exit.  Point fd 1 at stderr for the whole run and keep a private handle on the real
    stdout for the result line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    return real


def main():
    args = parse_args()
    real_stdout = _claim_stdout()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch N>1 with: python -m torch.distributed.run --nproc-per-node N ... '
                             'bench.py --gpus N (one process per GPU)')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback for the product path)'
    if args.comm == 'gloo':                      # ranks may share a GPU in this mode
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29531')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.comm == 'torch':
            dist.init_process_group('nccl', device_id=dev)   # "nccl" == RCCL on ROCm
        else:
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')   # single node: the hostname may not resolve
            dist.init_process_group('gloo')                  # bootstrap / barrier only; data goes over RCCL

    from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
    cof.load_library()

    N, H, C, K = args.batch, args.hw, args.channels, args.classes
    P = H * H
    tdtype = torch.float32 if args.dtype == 'f32' else torch.bfloat16
    g = torch.Generator(device='cpu').manual_seed(42 + rank)     # cfg.RNG_SEED = 42
    X = torch.randn(N, P, C, generator=g)
    X = (X if args.relu_input else torch.relu(X)).to(tdtype).to(dev)
    gw = torch.Generator(device='cpu').manual_seed(42)            # replicated weights
    Wa = (torch.randn(C, 1, generator=gw) / C ** 0.5).to(dev)
    ba = torch.zeros(1, device=dev)
    Wt = (torch.randn(C, K, generator=gw) / C ** 0.5).to(dev)
    bt = torch.zeros(K, device=dev)
    labels = torch.randint(0, K, (N,), generator=g).to(dev)

    train = not args.eval_mode
    flags = cof.attn_flags(args.softmax_att, False, train, relu_input=args.relu_input)
    keep = args.keep_prob if train else 1.0

    # flat fp32 gradient bucket [dWa | dba | dWt | dbt]: one all-reduce per step, no packing copy
    sizes = [C * 1, 1, C * K, K]
    bucket = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    views, o = [], 0
    for s in sizes:
        views.append(bucket[o:o + s])
        o += s
    dWa, dba, dWt, dbt = views[0].view(C, 1), views[1], views[2].view(C, K), views[3]
    dX = torch.empty_like(X)
    ws = torch.empty((cof.attn_pool_workspace_bytes(N, P, C, C, K, 1, flags),), dtype=torch.uint8,
                     device=dev)
    grad_scale = 1.0 / world                    # model_deploy.py:223-225: clone loss / num_clones
    # dropout step counter in HBM: read by the kernels, advanced by the backward call, so every
    # step (and every hipGraph replay) draws a fresh mask like a fresh tf.nn.dropout per sess.run
    rng_ctr = torch.zeros(1, dtype=torch.int64, device=dev)

    # one foreign call per step (apa_attn_head_train_step = the three entry points back to back,
    # arguments marshalled once): keeps the host ahead of the ~55 us step on any CPU
    stepper = None if args.per_op_calls else cof.HeadTrainStep(
        X, X, Wa, ba, Wt, bt, labels, (dX, None, dWa, dba, dWt, dbt), flags=flags, keep_prob=keep, seed=42,
        offset=rng_ctr, grad_scale=grad_scale, workspace=ws)

    def compute():
        if stepper is not None:
            stepper.run()
            return
        logits, att, zsave, abar, _, _ = cof.attn_pool_fwd(X, X, Wa, ba, Wt, bt, flags=flags,
                                                           keep_prob=keep, seed=42, offset=rng_ctr,
                                                           workspace=ws)
        _, G, _, _ = cof.softmax_xent_fwd_bwd(logits, labels, grad_scale=grad_scale)
        cof.attn_pool_bwd(X, X, Wa, ba, Wt, bt, att, zsave, abar, G, flags=flags, keep_prob=keep,
                          seed=42, offset=rng_ctr, workspace=ws,
                          out=(dX, None, dWa, dba, dWt, dbt))

    graph = None
    if args.graph:
        # the launches of a step are stream-ordered, allocation-free and argument-stable (the
        # dropout counter lives in HBM), so the whole step can be captured once in a hipGraph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                compute()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            compute()

    # Data-parallel gradient sum (model_deploy.py:421-451).  dWt|dbt (99.7 % of the payload) are final
    # after the first kernel of the backward call: the library records `ready` there and their
    # all-reduce starts on a side stream underneath the streaming pass; dWa|dba (8 KB) follow on the
    # main stream once the call is done.
    comm = comm_td = None
    if dist is not None and args.comm == 'rccl':
        from attentionalpoolingaction_amd import rccl
        ok = 1
        try:
            comm = rccl.RcclCommunicator(rank, max(world, 1), dev, group=None)
            comm.all_reduce_(torch.zeros(8, device=dev))      # first call builds the rings
            if args.overlap != 'off' and not args.graph:      # one communicator per stream
                comm_td = rccl.RcclCommunicator(rank, max(world, 1), dev, group=None)
                comm_td.all_reduce_(torch.zeros(8, device=dev))
            torch.cuda.synchronize()
        except Exception as e:                                 # noqa: BLE001 -- any failure -> fallback
            print('direct RCCL unavailable on rank {}: {}'.format(rank, e), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # every rank takes the same branch
        if int(flag.item()) == 0:                              # fall back to torch.distributed's RCCL
            for c in (comm, comm_td):
                if c is not None:
                    c.close()
            comm = comm_td = None
            args.comm = 'torch'
            dist.destroy_process_group()
            dist.init_process_group('nccl', device_id=dev)

    def allreduce(t, stream=None, async_op=False):
        if comm is not None:
            comm.all_reduce_(t, stream)
            return None
        return dist.all_reduce(t, async_op=async_op)

    overlap = None
    bucket_att, bucket_td = bucket[:C + 1], bucket[C + 1:]
    if dist is not None and graph is None and comm_td is not None:
        from attentionalpoolingaction_amd import deploy
        overlap = deploy.OverlappedGradientSum(bucket_att, bucket_td, comm, comm_td, dev)
    torch_overlap = dist is not None and graph is None and comm is None and args.overlap == 'on'

    def step(eager=False):
        if graph is not None and not eager:
            graph.replay()
        else:
            compute()
        if overlap is not None:
            overlap.after_backward()
        elif torch_overlap:
            w_td = allreduce(bucket_td, async_op=True)
            w_att = allreduce(bucket_att, async_op=True)
            w_td.wait()                         # the main stream waits for both (no host block)
            w_att.wait()
        elif dist is not None:
            allreduce(bucket)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0            # host time to enqueue the K steps (no device wait)
    barrier()
    elapsed = time.perf_counter() - t0
    print('host enqueue {:.1f} us/step, wall {:.1f} us/step'.format(t_enq / args.steps * 1e6,
                                                                 elapsed / args.steps * 1e6), file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant-kernel duration, HIP events on the launch stream, live over the same steps ----
    kt_steps = min(args.steps, 100)
    timer = cof.KernelTimer(kt_steps)
    for i in range(kt_steps):
        timer.arm(i)
        step(eager=True)        # event records are host calls: run the same launches un-captured
    timer.disarm()
    barrier()
    kms = sorted(timer.elapsed_ms())
    nms = sorted(timer.null_elapsed_ms())
    timer.close()
    # A HIP event pair costs ~3 us by itself (two timestamp packets + the dispatch gap); the library
    # records a second, empty pair right before the kernel's pair on the same stream, and the
    # kernel duration is the difference of the two averages.  rocprofv3 --kernel-trace (profiles/)
    # measures the same kernel begin->end on the GPU clock and must agree with it.
    k_raw_ms = sum(kms) / len(kms)
    k_null_ms = sum(nms) / len(nms)
    k_avg_ms = max(k_raw_ms - k_null_ms, 1e-6)
    esz = 4 if args.dtype == 'f32' else 2
    alg_bytes = 2.0 * N * P * C * esz           # bwd main kernel: read X once + write dX once
    achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel from the separate rocprofv3 --pmc passes (profiles/):
    # only valid for the exact workload it was collected on
    traffic = args.traffic_bytes
    workload = ('cfg002 attentional-pooling head fwd + softmax-xent + bwd, per-GPU batch {} '
                'x {}x{}x{} {} features, K={}, M=1 (class-agnostic bottom-up map), '
                '{}'.format(N, H, H, C, args.dtype, K,
                            'dropout keep={}'.format(keep) if train else 'eval (no dropout)'))
    if traffic is None and not args.softmax_att:
        # newest committed PMC summary whose workload string is exactly this run's
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
            try:
                pmc = json.load(open(f))
            except (OSError, ValueError):
                continue
            hit = [v for k, v in pmc.items() if 'bwd_main' in k and isinstance(v, dict)
                   and v.get('workload') == workload]
            if hit:
                traffic = hit[0]['hbm_bytes_per_launch']
                break

    if rank == 0:
        total_images = N * world * args.steps
        out = {
            'metric': 'images/sec attn-pool fwd+bwd, Nx14x14x2048, 393 classes',
            'value': round(total_images / elapsed, 1),
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {
                'workload': workload,
                'global_batch': N * world,
                'parallelism': 'dp{}'.format(world),
                'softmax_att': bool(args.softmax_att),
                'hip_graph': graph is not None,
                'host_calls_per_step': 3 if args.per_op_calls else 1,
                'comm': None if dist is None else ('librccl ncclAllReduce, in-stream' if comm is not None
                                                   else 'torch.distributed ' + dist.get_backend()),
                'allreduce': ('none' if dist is None else
                              'dWt|dbt on a communication stream (own communicator) between the grad-ready and '
                              'td-weights-ready hooks; dWa|dba in-stream' if overlap is not None else
                              'two async buckets' if torch_overlap else 'one bucket, in-stream'),
            },
            'roofline': {
                'bound': 'hbm',
                'kernel': 'm1s_bwd_main_kernel' if (C % 1024 == 0 and args.dtype == 'f32') or (C == 2048) else 'm1_bwd_main_kernel',
                'achieved': round(achieved, 1),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4),
                'traffic': traffic,
                'alg_bytes_per_launch': alg_bytes,
                'kernel_avg_us': round(k_avg_ms * 1e3, 3),
                'event_pair_raw_us': round(k_raw_ms * 1e3, 3),
                'event_pair_null_us': round(k_null_ms * 1e3, 3),
            },
            'step_roofline_frac': round((3.0 * N * P * C * esz) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
        }
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N = 1 only
            out['cpu_baseline'] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out), file=real_stdout, flush=True)
    if dist is not None:
        torch.cuda.synchronize()
        if overlap is not None:
            overlap.close()
        for c in (comm, comm_td):
            if c is not None:
                c.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
