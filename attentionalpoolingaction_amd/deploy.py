"""Data-parallel deployment of the head: one process per GPU, RCCL all-reduce over xGMI.

Reference semantics (/root/reference/models/slim/deployment/model_deploy.py):
  * every clone's loss is divided by num_clones          (_gather_clone_loss, :223-225)
  * per variable, the tower gradients are summed          (_sum_clones_gradients, :421-451)
  * the regularisation loss is added once (clone 0)       (optimize_clones, :294-309)
  * with TRAIN.ITER_SIZE > 1 gradients are accumulated locally and applied on the last
    micro-step                                           (src/train.py:529-566)
The reference does this in ONE process with towers on /gpu:i and the sum on the CPU device.  Here
each rank owns one MI355X and the sum is a single `all_reduce(SUM)` of one flat fp32 bucket
([dWa | dba | dWt | dbt | ...]: 3.2 MB for cfg 002) -- the kernels write their gradients straight
into views of that bucket, so there is no packing copy.  Backend "nccl" is RCCL on ROCm; the same
code runs on "gloo" with CPU tensors (tests/test_deploy_gloo_cpu.py, tests/test_train_reference_cpu.py,
world_size 2).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class DeploymentConfig:
    """The subset of slim's DeploymentConfig (model_deploy.py:481-683) that survives the move to
    one-process-per-GPU: the clone count and this process's clone index."""

    def __init__(self, num_clones: Optional[int] = None, clone_index: Optional[int] = None,
                 process_group=None):
        if dist.is_available() and dist.is_initialized():
            self.num_clones = dist.get_world_size(process_group) if num_clones is None else num_clones
            self.clone_index = dist.get_rank(process_group) if clone_index is None else clone_index
        else:
            self.num_clones = 1 if num_clones is None else num_clones
            self.clone_index = 0 if clone_index is None else clone_index
        self.process_group = process_group

    @property
    def clone_loss_scale(self) -> float:
        """model_deploy.py:223-225: clone_loss / num_clones (only when num_clones > 1)."""
        return 1.0 / self.num_clones if self.num_clones > 1 else 1.0

    def is_chief(self) -> bool:
        return self.clone_index == 0


class GradientBucket:
    """One flat fp32 buffer holding every head gradient; `views[name]` are the per-parameter
    windows the HIP kernels (or autograd) write into."""

    def __init__(self, shapes: Dict[str, Sequence[int]], device, dtype=torch.float32):
        self.names = list(shapes)
        self.shapes = {k: tuple(v) for k, v in shapes.items()}
        sizes = [int(torch.Size(self.shapes[k]).numel()) for k in self.names]
        self.flat = torch.zeros(sum(sizes), dtype=dtype, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        o = 0
        for k, s in zip(self.names, sizes):
            self.views[k] = self.flat[o:o + s].view(self.shapes[k])
            o += s

    @classmethod
    def for_parameters(cls, named_params: Iterable[Tuple[str, torch.Tensor]]):
        named = list(named_params)
        return cls({n: p.shape for n, p in named}, named[0][1].device)

    def zero_(self):
        self.flat.zero_()

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


def sum_clone_gradients(bucket: GradientBucket, config: DeploymentConfig, async_op: bool = False,
                        comm=None):
    """_sum_clones_gradients: one all-reduce(SUM) of the flat bucket.  The gradients must already
    carry the 1/num_clones loss scale (pass DeploymentConfig.clone_loss_scale as `grad_scale` to
    the loss kernel).  `comm`: an `rccl.RcclCommunicator` -> the collective is one in-stream
    ncclAllReduce on the current stream (no torch.distributed call on the step path); otherwise
    torch.distributed (gloo on CPU, "nccl" = RCCL on GPU), which returns the work handle when
    async_op=True (overlap with the next micro-batch's forward; wait before the optimizer step)."""
    if config.num_clones == 1:
        return None
    if comm is not None:
        comm.all_reduce_(bucket.flat)
        return None
    return dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=config.process_group,
                           async_op=async_op)


class HipRuntime:
    """streams / events / hooks of one GPU as OverlappedGradientSum uses them (torch.cuda + cof.make_hooks)"""

    def __init__(self, device):
        self.device = device

    def current_stream(self):
        return torch.cuda.current_stream(self.device)

    def new_stream(self):
        return torch.cuda.Stream(self.device)

    def new_event(self):
        return torch.cuda.Event()

    def synchronize(self) -> None:
        torch.cuda.synchronize(self.device)

    def stream(self, s):
        return torch.cuda.stream(s)

    def make_hooks(self, grad_ready, td_weights_ready):
        from .custom_ops import custom_ops_factory as cof
        return cof.make_hooks(grad_ready=grad_ready, td_weights_ready=td_weights_ready)


def probe_overlap_schedule(run_two_streams, run_in_stream, barrier, reduce_max, loops: int = 3, steps: int = 30):
    """`--overlap auto`: which gradient-sum schedule is faster on THIS node is measured, not assumed.  `loops` timed
    loops of `steps` steps of each schedule (after one warm loop each), the median loop of each MAX-reduced over the
    ranks -- every rank therefore sees the same two figures and takes the same decision, whatever its own clock
    said.  -> (keep_two_streams, us_two_streams, us_in_stream)."""
    import time

    def probe(fn):
        t = []
        for _ in range(loops):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            barrier()
            t.append((time.perf_counter() - t0) / steps * 1e6)
        return reduce_max(sorted(t)[len(t) // 2])
    probe(run_two_streams), probe(run_in_stream)               # (both warm)
    us_two, us_one = probe(run_two_streams), probe(run_in_stream)
    return us_two < us_one, us_two, us_one


class OverlappedGradientSum:
    """Two-stream gradient sum for the M == 1 head: the classifier part of the bucket (td_weights |
    td_biases, 99.7 % of the bytes) is final after the FIRST kernel of `apa_attn_pool_bwd` and is not
    read again before the logits product of the NEXT forward call, so its all-reduce (and, if given,
    its optimizer update) runs on a communication stream underneath the streaming backward pass of
    step k and the pooling / finalize passes of step k+1.  Only the attention part (att_weights |
    att_biases, 8 KB), produced by the last kernel of the backward call and needed by the first
    kernel of the next forward, is reduced on the compute stream.

        compute stream   fwd(k): pool, finalize, [wait td_done(k-1)] logits ... xent
                         bwd(k): head kernel -> record ready(k); streaming pass; column sums
                         all-reduce(att part)                          <- only exposed collective
        comm stream      [wait ready(k)] all-reduce(td part) -> update -> record td_done(k)

    The two waits are the `grad_ready_event` / `td_weights_ready_event` members of `apa_hooks`
    (include/apa.h): `self.hooks` is that struct and has to be handed to every forward / backward call
    of the head explicitly (`cof.attn_pool_fwd(..., hooks=ogs.hooks)`, `cof.attn_pool_bwd(..., hooks=
    ogs.hooks)`, `cof.HeadTrainStep(..., hooks=ogs.hooks)`; through the nn.Module surface:
    `network_fn.head.hooks = ogs.hooks`) -- nothing is registered per thread, so the
    schedule also holds when autograd runs the backward on its engine thread.  Each stream drives its own communicator: two collectives of one communicator
    must not be in flight at the same time.  Every rank enqueues them in the same order.  Same sums
    as `sum_clone_gradients` (model_deploy.py:421-451), only the schedule differs."""

    def __init__(self, bucket_att: torch.Tensor, bucket_td: torch.Tensor, comm_att, comm_td, device, runtime=None):
        """`runtime` (default: HIP through torch.cuda): where streams, events and the `apa_hooks` struct come from --
        injectable so that the schedule's bookkeeping (which communicator on which stream, the two event
        hand-overs, the order of the collectives) runs on two CPU ranks against stand-ins
        (tests/test_multi_rank_dryrun_cpu.py); see HipRuntime for the interface."""
        rt = HipRuntime(device) if runtime is None else runtime
        self.runtime = rt
        self.bucket_att, self.bucket_td = bucket_att, bucket_td
        self.comm_att, self.comm_td = comm_att, comm_td
        self.compute = rt.current_stream()
        self.side = rt.new_stream()
        self.ready, self.td_done = rt.new_event(), rt.new_event()
        self.ready.record(self.compute)          # materialise the handles; both start out signalled
        self.td_done.record(self.compute)
        rt.synchronize()
        self.hooks = rt.make_hooks(grad_ready=self.ready, td_weights_ready=self.td_done)

    def after_backward(self, update_td=None, update_att=None) -> None:
        """Call right after `attn_pool_bwd` was enqueued on the compute stream.  `update_td()` /
        `update_att()`: optional optimizer launches for the two parts; `update_td` runs with the
        communication stream current."""
        self.side.wait_event(self.ready)
        self.comm_td.all_reduce_(self.bucket_td, self.side)
        if update_td is not None:
            with self.runtime.stream(self.side):
                update_td()
        self.td_done.record(self.side)
        self.comm_att.all_reduce_(self.bucket_att, self.compute)
        if update_att is not None:
            update_att()

    def close(self) -> None:
        self.side.synchronize()


def add_regularization_gradient(bucket: GradientBucket, params: Dict[str, torch.Tensor],
                                weight_decay: float, regularized: Sequence[str]) -> None:
    """d/dW [ wd * 0.5 * |W|^2 ] = wd * W, added ONCE after the reduce -- equivalent to the
    reference adding the regularisation loss to clone 0 only (model_deploy.py:294-309)."""
    if weight_decay == 0.0:
        return
    for name in regularized:
        bucket.views[name].add_(params[name].detach().to(bucket.flat.dtype), alpha=weight_decay)


class GradientAccumulator:
    """TRAIN.ITER_SIZE semantics (src/train.py:529-566): micro-step gradients are summed into an
    accumulator, and only the last micro-step triggers reduce + apply; the accumulated gradient
    is divided by ITER_SIZE like train.py:556-560."""

    def __init__(self, bucket: GradientBucket, iter_size: int):
        self.bucket = bucket
        self.iter_size = max(1, int(iter_size))
        self.acc = torch.zeros_like(bucket.flat) if self.iter_size > 1 else None
        self._divisor = torch.full((), float(self.iter_size), dtype=bucket.flat.dtype, device=bucket.flat.device)
        self.micro = 0

    def step(self) -> bool:
        """Call after each micro-batch's backward.  Returns True when the bucket now holds the
        gradient to reduce and apply."""
        if self.iter_size == 1:
            return True
        self.acc.add_(self.bucket.flat)
        self.micro += 1
        if self.micro < self.iter_size:
            return False
        # `ref_grad / float(ITER_SIZE)` (src/train.py:560-563) is a true division.  torch divides a GPU tensor by
        # a python scalar as a multiplication by its reciprocal (an ulp off for ITER_SIZE = 3, 5, ...): divide by
        # a 0-dim tensor on the bucket's device instead, which takes the element-wise IEEE path
        torch.div(self.acc, self._divisor, out=self.bucket.flat)
        self.acc.zero_()
        self.micro = 0
        return True


class OverlappedMicroBatches:
    """The TRAIN.ITER_SIZE micro-batches of ONE update in flight side by side (round 3).

    The reference recipe for cfg 002 / 003 is `ITER_SIZE: 2` (experiments/002_MPII_ResNet_withAttention.yaml;
    src/train.py:529-566): two independent micro-batches, their gradients summed, one apply.  At the reference's
    per-GPU batch the head step is two HBM-streaming passes (~31 us) plus a chain of five latency-bound kernels
    (finalize -> logits -> xent -> backward head ... column sums, ~22 us on a handful of CUs: 42 % of the step,
    DESIGN.md section 3.3).  Here micro-batch i runs on its own HIP stream with its own workspace, dropout counter
    and gradient bucket; the compute stream then forms bucket = (g0 + g1 + ...) / ITER_SIZE in one launch.

    What it buys, measured on MI355X (N = 32 per micro-batch, fp32, bench.py extra.cfg002_train_iter_size): the
    lanes run nearly in lockstep -- pooling beside pooling, chain beside chain, backward beside backward (device
    timeline from the dispatch events) -- so what overlaps is one chain with the other:
    99-102 us per update against 109-110 us back to back (-8 %), per-image step fraction 0.354 -> 0.385.  The
    hoped-for pairing "streaming pass of B underneath the chain of A" (forced with a cross-stream event after A's
    pooling pass, eagerly and inside a hipGraph) measured WORSE (110-118 us): the chain kernels are bound by
    memory latency, and beside a kernel that saturates HBM their loads take about twice as long, while two
    event operations per lane cost the host 12 us.  That variant was removed again.  On this chip the way to
    amortise the chain is a larger pass, not concurrency: the same two micro-batches as ONE call over 2N images
    (memory is no constraint with 288 GB) take 80 us, 0.48 of the HBM roofline (bench.py reports it next to these figures).

    Same kernels, same inputs, a fixed summation order: the result is bit-identical to running the micro-batches
    one after the other through `GradientAccumulator` (tests/test_head_gpu.py).  `steppers[i]` is the
    `cof.HeadTrainStep` of lane i (its gradient views must point into `lane_buckets[i]`)."""

    def __init__(self, steppers: Sequence, lane_buckets: Sequence[torch.Tensor], out_bucket: torch.Tensor, device):
        from .custom_ops import custom_ops_factory as cof
        assert len(steppers) == len(lane_buckets) >= 1
        self._cof = cof
        self.steppers = list(steppers)
        self.lane_buckets = list(lane_buckets)
        self.out = out_bucket
        self.main = torch.cuda.current_stream(device)
        self.sides = [torch.cuda.Stream(device) for _ in steppers[1:]]
        self.start = torch.cuda.Event()
        self.done = [torch.cuda.Event() for _ in steppers[1:]]

    def run(self, steppers: Optional[Sequence] = None) -> None:
        """Enqueue one update's worth of micro-batches.  `steppers` overrides the lanes' steppers for this call
        (e.g. rotating input buffers); every lane keeps its own stream."""
        st = self.steppers if steppers is None else steppers
        self.start.record(self.main)                 # the lanes start after whatever precedes on the compute stream
        for side in self.sides:                      # (the optimizer update of the previous iteration)
            side.wait_event(self.start)
        st[0].run(stream=self.main.cuda_stream)
        for s_, side, ev in zip(st[1:], self.sides, self.done):
            s_.run(stream=side.cuda_stream)
            ev.record(side)
        for ev in self.done:
            self.main.wait_event(ev)
        self._cof.accumulate_gradients(self.out, self.lane_buckets, divisor=float(len(st)),
                                       stream=self.main.cuda_stream)

    def run_sequential(self, steppers: Optional[Sequence] = None) -> None:
        """The same update with the micro-batches one after the other on the compute stream (the schedule of
        `GradientAccumulator`): the yardstick the overlapped form is measured and checked against."""
        st = self.steppers if steppers is None else steppers
        for s_ in st:
            s_.run(stream=self.main.cuda_stream)
        self._cof.accumulate_gradients(self.out, self.lane_buckets, divisor=float(len(st)),
                                       stream=self.main.cuda_stream)

    def close(self) -> None:
        for side in self.sides:
            side.synchronize()


class StaleOperandError(RuntimeError):
    """A head parameter was written behind the optimiser's back (load_state_dict, a checkpoint restore, any in-place
    op on the Parameter) while a bf16 shadow / operand image of it is in use: the next step would run on the OLD
    weights.  Call `optimizer.refresh_shadows()` (FusedHeadStep: `refresh_operands()`), or build the optimiser with
    `stale='refresh'`."""


class MomentumSGD:
    """tf.train.MomentumOptimizer(lr, momentum) (src/train.py:90-94): acc = m*acc + g;
    w -= lr*acc (no Nesterov, no dampening) on the flat bucket layout, with the slim L2
    regulariser's gradient `wd * w` folded in for the names in `regularized`
    (resnet_utils.py:241: conv weights only).  On a GPU the whole update is ONE fused HIP launch
    (`apa_momentum_sgd_step`); CPU tensors (the gloo tests) take the equivalent torch expressions.

    Operand copies kept current by the update (round 4-5): bf16 shadows of whole parameters, and "images" (affine
    scatter maps into a step's workspace, `cof.ApaWeightImage`).  Every image entry carries its OWNER -- the object
    whose memory `map.dst` points into -- so the block cannot return to the allocator while the launch still
    scatters into it, and `detach_weight_images(owner)` drops it again.

    Staleness guard (round 6): `params` may be the nn.Parameters themselves.  The fused launches write through raw
    pointers / `.data` and never bump `Parameter._version`; anything else that writes a weight (load_state_dict,
    `with torch.no_grad(): p.copy_(...)`) does.  The version of every parameter that has a shadow or an image is
    recorded whenever its copies are (re)written and compared in `check_fresh()` -- called by `step()` and by
    `FusedHeadStep` before each head step: `stale='raise'` (default) -> StaleOperandError, `'refresh'` -> the copies
    are rebuilt on the spot."""

    fused_images = True         # the launch of this optimiser carries image maps (the adaptive ones do not)

    def __init__(self, params: Dict[str, torch.Tensor], bucket: GradientBucket, lr: float,
                 momentum: float = 0.9, weight_decay: float = 0.0, regularized: Sequence[str] = (),
                 bf16_shadows: Optional[Dict[str, torch.Tensor]] = None, stale: str = 'raise'):
        """`bf16_shadows` {name: bf16 tensor}: operand copies the bf16 MFMA products read (the pose head's W1 for
        `cof.PoseAttnTrainStep(w1_bf16=...)`), rewritten from the updated weights by the update's own launch."""
        if stale not in ('raise', 'refresh'):
            raise ValueError("stale: 'raise' or 'refresh'")
        self.stale = stale
        self.shadows = dict(bf16_shadows or {})
        self.params = params
        self.bucket = bucket
        self._bound = None          # cof.BoundMomentumSGD of the current shadow / image set (marshalled once)
        self.images = []            # [(segment, ApaWeightImage, owner)]: rewritten by the update's own launch
        self._img_refresh = []      # owners rebuilt by their own `refresh_weight_images()` after each update
        self._seen = {}             # name -> Parameter._version when its shadow / images were last written
        for name, t in self.shadows.items():
            if name not in params or t.dtype != torch.bfloat16 or t.numel() != params[name].numel():
                raise ValueError('bf16 shadow %r: a bfloat16 tensor with the element count of that parameter' % name)
        self.refresh_shadows()          # a shadow is current from the start, not only after the first update
        self.lr = lr
        self.momentum = momentum
        self.acc = torch.zeros_like(bucket.flat)
        reg = set(regularized)
        self.wd = [weight_decay if n in reg else 0.0 for n in bucket.names]

    # -- operand images ------------------------------------------------------------------------------------------
    def attach_weight_images(self, step, names: Dict[str, str]) -> None:
        """Keep the per-class head's operand images current in THIS optimiser's launch: `step` is a
        cof.HeadTrainStep(..., weight_images=True) (or anything with `.weight_image_maps` and
        `.refresh_weight_images()`), `names` maps the roles 'Wa' / 'ba' / 'Wt' / 'bt' to parameter names of the
        bucket.  Several steps (micro-batch lanes, rotating buffer sets) may be attached; at most three images per
        parameter fit the fused launch, further ones are rebuilt by their step's own refresh launch after the
        update.  The optimiser keeps `step` (hence its workspace) alive until `detach_weight_images(step)`."""
        watched = [names[role] for role, _ in step.weight_image_maps]
        count = {}
        for seg, _, _ in self.images:
            count[seg] = count.get(seg, 0) + 1
        fits = self.fused_images
        new = []
        for role, m in step.weight_image_maps:
            seg = self.bucket.names.index(names[role])
            count[seg] = count.get(seg, 0) + 1
            fits = fits and count[seg] <= 3
            new.append((seg, m, step))
        if fits:
            self.images += new
        else:
            self._img_refresh.append(step)
        self._bound = None
        step._image_param_names = watched
        self._note_written(watched)

    def detach_weight_images(self, owner) -> None:
        """Forget every image that lives in `owner`'s memory (a step that is dropped or re-bound): the launch stops
        scattering into it and the reference that kept it alive is released."""
        self.images = [e for e in self.images if e[2] is not owner]
        self._img_refresh = [o for o in self._img_refresh if o is not owner]
        self._bound = None

    def add_image(self, name: str, image_map, refresh=None, owner=None) -> None:
        """One more operand image of parameter `name` (a cof.ApaWeightImage, e.g. cof.pose_w2t_image_map: the bf16
        transposed copy of the pose head's W2) to be rewritten by this optimiser's launch; `refresh()` rebuilds it
        from the weights where the launch cannot (more than three images on one parameter; the adaptive optimisers,
        whose launches carry no image maps) and in `refresh_images()`; `owner`: the tensor / object `image_map.dst`
        points into (kept alive here)."""
        seg = self.bucket.names.index(name)
        holder = _ImageRefresh(refresh, owner, [name])
        if self.fused_images and sum(1 for s_, _, _ in self.images if s_ == seg) < 3:
            self.images.append((seg, image_map, holder))
        elif refresh is not None:
            self._img_refresh.append(holder)
        else:
            raise ValueError('add_image(%r): this optimiser cannot rewrite the image in its launch; pass refresh=' % name)
        self._bound = None
        self._note_written([name])

    def _image_owners(self):
        seen, out = set(), []
        for o in [e[2] for e in self.images] + list(self._img_refresh):
            if id(o) not in seen:
                seen.add(id(o))
                out.append(o)
        return out

    def refresh_images(self) -> None:
        """Rebuild EVERY attached operand image from the current weights (each owner's own refresh launch)."""
        for o in self._image_owners():
            o.refresh_weight_images()
            self._note_written(getattr(o, '_image_param_names', ()))

    def refresh_shadows(self) -> None:
        """Rewrite every bf16 operand copy AND every attached operand image from its parameter: at construction, and
        after the weights were changed behind the optimiser's back (load_state_dict, a checkpoint restore)."""
        with torch.no_grad():
            for name, t in self.shadows.items():
                t.copy_(self.params[name].data.reshape(t.shape))
        self._note_written(self.shadows)
        self.refresh_images()

    # -- staleness guard -----------------------------------------------------------------------------------------
    def _note_written(self, names) -> None:
        for n in names:
            self._seen[n] = getattr(self.params[n], '_version', 0)

    def stale_names(self):
        return [n for n, v in self._seen.items() if getattr(self.params[n], '_version', 0) != v]

    def check_fresh(self) -> None:
        """Compare `Parameter._version` of every parameter with a shadow / image against the value recorded when
        the copies were last written (a host-side integer compare, no device work)."""
        bad = self.stale_names()
        if not bad:
            return
        if self.stale == 'refresh':
            self.refresh_shadows()
            return
        raise StaleOperandError(
            '%s written outside the optimiser since their bf16 shadow / operand images were built: the head would '
            'run on the old weights -- call refresh_shadows() after load_state_dict / a restore' % ', '.join(bad))

    def _after_update(self) -> None:
        for o in self._img_refresh:
            o.refresh_weight_images()

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        lr = self.lr if lr is None else lr
        self.check_fresh()
        if self.bucket.flat.is_cuda:
            if self._bound is None:                      # marshalled once per shadow / image set
                from .custom_ops import custom_ops_factory as cof
                ws = [self.params[n].data for n in self.bucket.names]
                sh = [self.shadows.get(n) for n in self.bucket.names] if self.shadows else None
                self._bound = cof.BoundMomentumSGD(ws, self.wd, self.bucket.flat, self.acc, shadows=sh,
                                                   images=[(seg, m) for seg, m, _ in self.images] or None)
            self._bound.run(lr, self.momentum, grad_scale)
            self._after_update()
            return
        ws = [self.params[n].data for n in self.bucket.names]
        o = 0
        for w, wd in zip(ws, self.wd):
            n = w.numel()
            g = self.bucket.flat[o:o + n].view_as(w) * grad_scale + wd * w
            a = self.acc[o:o + n].view_as(w)
            a.mul_(self.momentum).add_(g)
            w.add_(a, alpha=-lr)
            o += n
        for name, t in self.shadows.items():
            t.copy_(self.params[name].data.reshape(t.shape))


class _ImageRefresh:
    """owner record of one `add_image` entry: keeps the image's memory alive and knows how to rebuild it"""

    def __init__(self, refresh, owner, names):
        self._refresh, self.owner, self._image_param_names = refresh, owner, list(names)

    def refresh_weight_images(self) -> None:
        if self._refresh is not None:
            self._refresh()


class _AdaptiveOptimizer(MomentumSGD):
    """shared plumbing of Adam / RMSProp below: two slot buffers in the bucket's layout, the L2 term folded in"""

    fused_images = False        # the adaptive launches carry no image maps: an attached owner rebuilds its images
                                # by its own refresh launch after each update (MomentumSGD._after_update)

    def __init__(self, params, bucket, lr, weight_decay=0.0, regularized=(), bf16_shadows=None, stale='raise'):
        super().__init__(params, bucket, lr, 0.0, weight_decay, regularized, bf16_shadows, stale)
        self.slot2 = torch.zeros_like(bucket.flat)          # self.acc is the first slot

    def _cpu_shadows(self) -> None:
        with torch.no_grad():
            for name, t in self.shadows.items():
                t.copy_(self.params[name].data.reshape(t.shape))

    def _segments(self, grad_scale):
        o = 0
        for name, wd in zip(self.bucket.names, self.wd):
            w = self.params[name].data
            n = w.numel()
            g = self.bucket.flat[o:o + n].view_as(w) * grad_scale + wd * w
            yield w, g, self.acc[o:o + n].view_as(w), self.slot2[o:o + n].view_as(w)
            o += n


class Adam(_AdaptiveOptimizer):
    """tf.train.AdamOptimizer(lr, beta1, beta2, epsilon) (src/train.py:84-89; TF 1.1 training/adam.py):
    lr_t = lr sqrt(1 - b2^t)/(1 - b1^t);  m += (g - m)(1 - b1);  v += (g^2 - v)(1 - b2);  w -= lr_t m/(sqrt(v) + eps).
    Note the reference's epsilon: cfg.TRAIN.OPT_EPSILON defaults to 1.0 (src/config.py:94).  One fused HIP launch
    (`apa_adam_step`) on a GPU; the torch expressions on CPU tensors."""

    def __init__(self, params, bucket, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.0, regularized=(),
                 bf16_shadows=None, stale='raise'):
        super().__init__(params, bucket, lr, weight_decay, regularized, bf16_shadows, stale)
        self.beta1, self.beta2, self.epsilon, self.t = float(beta1), float(beta2), float(epsilon), 0

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        lr = self.lr if lr is None else lr
        self.check_fresh()
        self.t += 1
        if self.bucket.flat.is_cuda:
            from .custom_ops import custom_ops_factory as cof
            ws = [self.params[n].data for n in self.bucket.names]
            sh = [self.shadows.get(n) for n in self.bucket.names] if self.shadows else None
            cof.adam_step(ws, self.wd, self.bucket.flat, self.acc, self.slot2, lr, self.t, self.beta1, self.beta2,
                          self.epsilon, grad_scale, shadows=sh)
            self._after_update()
            return
        lr_t = lr * (1.0 - self.beta2 ** self.t) ** 0.5 / (1.0 - self.beta1 ** self.t)
        for w, g, m, v in self._segments(grad_scale):
            m.add_((g - m) * (1.0 - self.beta1))
            v.add_((g * g - v) * (1.0 - self.beta2))
            w.sub_(m * lr_t / (v.sqrt() + self.epsilon))
        self._cpu_shadows()


class RMSProp(_AdaptiveOptimizer):
    """tf.train.RMSPropOptimizer(lr, decay, momentum, epsilon) (src/train.py:95-100; TF 1.1 training/rmsprop.py, not
    centered):  ms += (g^2 - ms)(1 - decay);  mom = momentum mom + lr g / sqrt(ms + eps);  w -= mom;  `ms` starts at
    ONE.  One fused HIP launch (`apa_rmsprop_step`) on a GPU."""

    def __init__(self, params, bucket, lr, decay=0.9, momentum=0.0, epsilon=1e-10, weight_decay=0.0, regularized=(),
                 bf16_shadows=None, stale='raise'):
        super().__init__(params, bucket, lr, weight_decay, regularized, bf16_shadows, stale)
        self.decay, self.momentum, self.epsilon = float(decay), float(momentum), float(epsilon)
        self.acc.fill_(1.0)                                  # rmsprop.py _create_slots: init_rms = ones

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        lr = self.lr if lr is None else lr
        self.check_fresh()
        if self.bucket.flat.is_cuda:
            from .custom_ops import custom_ops_factory as cof
            ws = [self.params[n].data for n in self.bucket.names]
            sh = [self.shadows.get(n) for n in self.bucket.names] if self.shadows else None
            cof.rmsprop_step(ws, self.wd, self.bucket.flat, self.acc, self.slot2, lr, self.decay, self.momentum,
                             self.epsilon, grad_scale, shadows=sh)
            self._after_update()
            return
        for w, g, ms, mom in self._segments(grad_scale):
            ms.add_((g * g - ms) * (1.0 - self.decay))
            mom.mul_(self.momentum).add_(g * lr / (ms + self.epsilon).sqrt())
            w.sub_(mom)
        self._cpu_shadows()


def exponential_decay_lr(base_lr: float, global_step: int, decay_steps: int, decay_rate: float,
                         staircase: bool = True) -> float:
    """tf.train.exponential_decay (src/train.py:50-56): lr * rate^floor(step/decay_steps)."""
    e = global_step / float(decay_steps)
    if staircase:
        e = float(int(e))
    return base_lr * (decay_rate ** e)


def decay_steps(cfg, num_samples_per_epoch: int, num_clones: int) -> int:
    """The step count between two learning-rate drops, src/train.py:42-48: TRAIN.NUM_STEPS_PER_DECAY when it
    is positive, else int(samples / (BATCH_SIZE * clones * ITER_SIZE) * NUM_EPOCHS_PER_DECAY) -- one "step"
    is one parameter update, i.e. ITER_SIZE micro-batches on every clone."""
    if cfg.TRAIN.NUM_STEPS_PER_DECAY > 0:
        return int(cfg.TRAIN.NUM_STEPS_PER_DECAY)
    return int(num_samples_per_epoch / (cfg.TRAIN.BATCH_SIZE * num_clones * cfg.TRAIN.ITER_SIZE)
               * cfg.TRAIN.NUM_EPOCHS_PER_DECAY)


def configure_learning_rate(cfg, num_samples_per_epoch: int, num_clones: int, global_step: int) -> float:
    """_configure_learning_rate (src/train.py:29-69) evaluated at `global_step` (the number of parameter
    updates applied so far).  'exponential' is the staircase decay, 'fixed' the constant rate; the reference's
    'polynomial' branch (power 1, no cycle) is the linear ramp to END_LEARNING_RATE."""
    kind = cfg.TRAIN.LEARNING_RATE_DECAY_TYPE
    if kind == 'exponential':
        return exponential_decay_lr(cfg.TRAIN.LEARNING_RATE, global_step,
                                    decay_steps(cfg, num_samples_per_epoch, num_clones),
                                    cfg.TRAIN.LEARNING_RATE_DECAY_RATE, staircase=True)
    if kind == 'fixed':
        return float(cfg.TRAIN.LEARNING_RATE)
    if kind == 'polynomial':
        n = decay_steps(cfg, num_samples_per_epoch, num_clones)
        t = min(global_step, n) / float(n)
        return (cfg.TRAIN.LEARNING_RATE - cfg.TRAIN.END_LEARNING_RATE) * (1.0 - t) + cfg.TRAIN.END_LEARNING_RATE
    raise ValueError('learning_rate_decay_type [%s] was not recognized' % kind)


def configure_optimizer(cfg, params: Dict[str, torch.Tensor], bucket: GradientBucket, learning_rate: float,
                        regularized: Sequence[str] = (), bf16_shadows: Optional[Dict[str, torch.Tensor]] = None,
                        stale: str = 'raise'):
    """_configure_optimizer (src/train.py:72-105): every optimiser the reference can select, each as ONE fused
    launch -- 'momentum' (cfgs 001-003: MomentumOptimizer(lr, TRAIN.MOMENTUM)), 'sgd' (momentum 0), 'adam'
    (TRAIN.ADAM_BETA1 / ADAM_BETA2 / OPT_EPSILON) and 'rmsprop' (TRAIN.RMSPROP_DECAY / MOMENTUM / OPT_EPSILON).  The
    L2 regulariser's gradient (TRAIN.WEIGHT_DECAY on `regularized`) is folded into the same launch.  As in the
    reference, 'rmsprop' reads cfg.TRAIN.RMSPROP_DECAY, a key src/config.py does not define: without it in the
    YAML the reference fails with an AttributeError at this point, and so does this function."""
    kind = cfg.TRAIN.OPTIMIZER
    wd = float(cfg.TRAIN.WEIGHT_DECAY)
    if kind == 'adam':
        return Adam(params, bucket, lr=learning_rate, beta1=float(cfg.TRAIN.ADAM_BETA1),
                    beta2=float(cfg.TRAIN.ADAM_BETA2), epsilon=float(cfg.TRAIN.OPT_EPSILON), weight_decay=wd,
                    regularized=regularized, bf16_shadows=bf16_shadows, stale=stale)
    if kind == 'rmsprop':
        if 'RMSPROP_DECAY' not in cfg.TRAIN:
            raise AttributeError('RMSPROP_DECAY')            # src/train.py:98 on the reference's own config table
        return RMSProp(params, bucket, lr=learning_rate, decay=float(cfg.TRAIN.RMSPROP_DECAY),
                       momentum=float(cfg.TRAIN.MOMENTUM), epsilon=float(cfg.TRAIN.OPT_EPSILON), weight_decay=wd,
                       regularized=regularized, bf16_shadows=bf16_shadows, stale=stale)
    if kind == 'momentum':
        momentum = float(cfg.TRAIN.MOMENTUM)
    elif kind == 'sgd':
        momentum = 0.0
    else:
        raise ValueError('Optimizer [%s] was not recognized' % kind)
    return MomentumSGD(params, bucket, lr=learning_rate, momentum=momentum,
                       weight_decay=float(cfg.TRAIN.WEIGHT_DECAY), regularized=regularized, bf16_shadows=bf16_shadows,
                       stale=stale)


class _FusedHeadFunction(torch.autograd.Function):
    """autograd node of FusedHeadStep: the forward IS the whole head step (forward + losses + backward, one host
    call); the backward hands the stored conv5 gradient on to the backbone.  `anchor` is a 0-dim leaf that requires
    grad: with it the node exists (and scales the bucket by the upstream coefficient) also when nothing upstream of
    conv5 is differentiated -- head-only training takes the same path as the end-to-end step."""

    @staticmethod
    def forward(ctx, last_conv, anchor, owner, labels_action, labels_pose, pose_valid):
        total = owner._run(last_conv, labels_action, labels_pose, pose_valid)
        ctx.owner = owner
        ctx.step = owner._step_obj
        ctx.xshape = last_conv.shape
        return total

    @staticmethod
    def backward(ctx, g_total):
        o = ctx.owner
        dX = ctx.step._dX_buf.view(ctx.xshape) if ctx.needs_input_grad[0] else None
        if not o.assume_unit_upstream:        # total entered the differentiated scalar with some coefficient
            if dX is not None:
                dX = dX * g_total.to(dX.dtype)
            for n in o._written:
                o.bucket.views[n].mul_(g_total)
        return dX, None, None, None, None, None


class FusedHeadStep:
    """The reference's per-clone training graph for the head -- `clone_fn` (network_fn -> gen_losses,
    src/train.py:393-422) plus `optimizer.compute_gradients` on the clone loss (model_deploy.py:263) -- as ONE host
    call on the conv5 map: `apa_pose_attn_train_step` for the cfg 003 form (attention from pose_pre_logits, pose L2 +
    softmax cross-entropy), `apa_attn_head_train_step` for the cfg 002 / per-class (HMDB-51) forms (attention from
    the map itself, softmax cross-entropy).  The YAML-driven surface stays the reference's: `get_network_fn(...)`
    builds backbone + head, this object drives them.

        fused = deploy.FusedHeadStep(network_fn, cfg)                 # raises ValueError if the configuration
        opt = fused.make_optimizer(lr)                                #   needs the per-op module path
        total, end_points = fused(images, labels_action, labels_pose, pose_valid)
        total.backward()              # conv5's gradient enters the backbone's autograd graph; the head's gradients
        opt.step()                    # already lie in fused.bucket (the flat all-reduce payload)

    * `total` = sum of the clone's tf.losses entries (each scaled by `loss_scale` = 1 / num_clones,
      model_deploy.py:223-225); the L2 regulariser is the optimiser's (`weight_decay * w` folded into its launch).
      end_points: 'Logits', 'PosePrelogitsBasedAttention', 'PoseLogits' (cfg 003), 'Losses' (the individual values
      in tf.GraphKeys.LOSSES order, detached COPIES; the other end points are views of the bound step's buffers,
      overwritten by the next step of the same batch shape).
    * the head's gradients are written straight into `bucket` views, in the step's own launches; the bf16 operand
      copy of the pose head's W1 is owned by the optimiser (`make_optimizer`: `bf16_shadows`) and rewritten by its
      update launch, so no conversion kernel runs in the step.
    * `assume_unit_upstream=True`: the caller promises `total` enters the differentiated scalar with coefficient 1
      (`(total + other).backward()`), which saves one pass over the [N,H,W,C] gradient; the default multiplies by
      whatever arrives -- conv5's gradient and the bucket alike, with or without a differentiated backbone.
    * one bound step (outputs, workspace, conv5-gradient buffer) is kept per batch shape / dtype (`max_bound_steps`
      most recent: a smaller last batch of an epoch comes back every epoch).  Only the step that is about to run
      has its operand images attached to the optimiser's launch; switching detaches the previous one (the launch
      must not scatter into a workspace nobody vouches for any more) and rebuilds the images of the incoming one.
    * before every step the optimiser's staleness guard runs (`MomentumSGD.check_fresh`): a weight written behind its
      back (load_state_dict after make_optimizer) raises StaleOperandError -- or, with `stale='refresh'`, rebuilds
      the bf16 shadow and every operand image first.  `refresh_operands()` does that by hand."""

    max_bound_steps = 4

    def __init__(self, network_fn, cfg, loss_scale: float = 1.0, assume_unit_upstream: bool = False,
                 stale: str = 'raise'):
        from . import nets_factory
        head = network_fn.head
        if not isinstance(head, nets_factory.AttentionalPoolingHead):
            raise ValueError('FusedHeadStep: the attentional-pooling head only (cfg 001 has no attention op)')
        why = self.unsupported_reason(head, cfg, network_fn)
        if why:
            raise ValueError('FusedHeadStep: ' + why)
        self.network_fn, self.head, self.cfg = network_fn, head, cfg
        self.loss_scale = float(loss_scale)
        self.assume_unit_upstream = bool(assume_unit_upstream)
        self.stale = stale
        self.pose_form = not head.single_layer
        names = ['pose_w1', 'pose_b1', 'pose_w2', 'pose_b2', 'att_weights', 'att_biases', 'td_weights', 'td_biases']
        self.params = {n: getattr(head, n) for n in names}
        self.bucket = GradientBucket.for_parameters(self.params.items())
        # cfg 002 forms: the PoseLogits convs are pruned from the data path -- no gradient, but their weights are
        # regularised and decay (the optimiser's L2 term on a zero gradient), as in the reference (tf.gradients
        # reaches them through REGULARIZATION_LOSSES only)
        self._written = names if self.pose_form else names[4:]
        reg = {id(w) for w in head.regularized_weights()}
        self.regularized = [n for n in names if id(self.params[n]) in reg]
        self._steps = {}                 # key (shape, dtype, preact) -> bound one-call step, most recent last
        self._step_obj = None
        self._key = None
        self.w1_shadow = None
        self.w2t_image = None
        self._optimizer = None
        self._anchor = torch.zeros((), requires_grad=True)
        self.probe_events = None

    @staticmethod
    def unsupported_reason(head, cfg, network_fn=None) -> str:
        tr = cfg.TRAIN
        if not head.is_training:
            return 'a training-mode head is required'
        if head.rank != 1 or head.with_pose_feat or head.want_topdown:
            return 'rank > 1, ..._WITH_POSE_FEAT and the TopDownAttention dump run through the per-op module'
        if tr.LOSS_FN_ACTION != 'softmax-xentropy':
            return 'LOSS_FN_ACTION %r (the one-call steps take the softmax cross-entropy)' % tr.LOSS_FN_ACTION
        if network_fn is not None and network_fn.temporal is not None:
            return 'temporal attention / frame pooling sit between the head and the loss'
        if head.single_layer:
            if tr.LOSS_FN_POSE and head.with_pose_logits:
                return 'a pose loss on a head whose attention does not come from the pose head'
        else:
            if head.per_class:
                return 'per-class maps from pose_pre_logits'
            if tr.LOSS_FN_POSE != 'l2' or tr.LOSS_FN_POSE_SAMPLED:
                return 'LOSS_FN_POSE %r / sampled (the cfg 003 step takes the plain pose L2 loss)' % tr.LOSS_FN_POSE
        return ''

    @property
    def _dX(self):
        return None if self._step_obj is None else self._step_obj._dX_buf

    def _drop_steps(self) -> None:
        if self._optimizer is not None:
            for st in self._steps.values():
                self._optimizer.detach_weight_images(st)
        self._steps.clear()
        self._step_obj = self._key = None

    def make_optimizer(self, learning_rate: float):
        """deploy.configure_optimizer on the head's parameters (the nn.Parameters themselves: their `_version` feeds
        the staleness guard) and this object's bucket, with the bf16 copy of the pose head's W1 as a shadow of the
        update launch when the cfg 003 step will read one (bf16 features)."""
        shadows = None
        self._drop_steps()                           # re-bind with the shadow / the weight images
        if self.pose_form:
            w1 = self.params['pose_w1']
            self.w1_shadow = torch.empty(w1.shape, dtype=torch.bfloat16, device=w1.device)
            shadows = {'pose_w1': self.w1_shadow}
        self._optimizer = configure_optimizer(self.cfg, dict(self.params), self.bucket, learning_rate,
                                              regularized=self.regularized, bf16_shadows=shadows, stale=self.stale)
        w2 = self.params['pose_w2'].data
        if self.pose_form and w2.is_cuda and w2.shape[1] <= 16:
            # the Pl product reads W2^T as a ready-made bf16 image, rewritten by the optimiser's launch as well
            from .custom_ops import custom_ops_factory as cof
            self.w2t_image = cof.pose_w2t_image(w2)
            img = self.w2t_image
            self._optimizer.add_image('pose_w2', cof.pose_w2t_image_map(img, w2), owner=img,
                                      refresh=lambda: img[:w2.shape[1], :w2.shape[0]].copy_(w2.t()))
        return self._optimizer

    def refresh_operands(self) -> None:
        """Rebuild the bf16 W1 shadow, the W2^T image and the per-class weight images of the bound step from the
        current weights (after load_state_dict / a checkpoint restore that happened after `make_optimizer`)."""
        if self._optimizer is not None:
            self._optimizer.refresh_shadows()
        elif self._step_obj is not None and getattr(self._step_obj, 'weight_image_maps', None):
            self._step_obj.refresh_weight_images()

    _IMAGE_ROLES = {'Wa': 'att_weights', 'ba': 'att_biases', 'Wt': 'td_weights', 'bt': 'td_biases'}

    def _bind(self, X, labels_action, labels_pose, pose_valid):
        from .custom_ops import custom_ops_factory as cof
        head, tr, v = self.head, self.cfg.TRAIN, self.bucket.views
        flags = cof.attn_flags(head.softmax_att, head.relu_att, True, self._preact)
        dX = torch.empty_like(X)
        p = {n: t.data for n, t in self.params.items()}
        if self.pose_form:
            shadow = self.w1_shadow if X.dtype == torch.bfloat16 else None
            st = cof.PoseAttnTrainStep(
                X, (p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'], p['att_weights'], p['att_biases'],
                    p['td_weights'], p['td_biases']), labels_action, labels_pose, pose_valid,
                (dX, v['pose_w1'], v['pose_b1'], v['pose_w2'], v['pose_b2'], v['att_weights'], v['att_biases'],
                 v['td_weights'], v['td_biases']), flags=flags, keep_prob=head.keep_prob, seed=head.seed,
                offset=head._step, action_wt=float(tr.LOSS_FN_ACTION_WT), pose_wt=float(tr.LOSS_FN_POSE_WT),
                grad_scale=self.loss_scale, w1_bf16=shadow,
                w2t_bf16=self.w2t_image if X.dtype == torch.bfloat16 else None)
        else:
            st = cof.HeadTrainStep(
                X, X, p['att_weights'], p['att_biases'], p['td_weights'], p['td_biases'], labels_action,
                (dX, None, v['att_weights'], v['att_biases'], v['td_weights'], v['td_biases']), flags=flags,
                keep_prob=head.keep_prob, seed=head.seed, offset=head._step, loss_wt=float(tr.LOSS_FN_ACTION_WT),
                grad_scale=self.loss_scale, hooks=head.hooks,
                weight_images=bool(head.per_class and self._optimizer is not None))
        st._dX_buf = dX
        return st

    def _select(self, key, X, labels_action, labels_pose, pose_valid):
        """make the bound step of `key` the current one: at most one step has its operand images attached"""
        prev = self._step_obj
        st = self._steps.pop(key, None)
        cached = st is not None
        if st is None:
            st = self._bind(X, labels_action, labels_pose, pose_valid)
        self._steps[key] = st                            # most recent last
        while len(self._steps) > self.max_bound_steps:
            old_key = next(iter(self._steps))
            old = self._steps.pop(old_key)
            if self._optimizer is not None:
                self._optimizer.detach_weight_images(old)
        if st is not prev and self._optimizer is not None:
            if prev is not None:
                self._optimizer.detach_weight_images(prev)
            if getattr(st, 'weight_image_maps', None):   # per-class maps: rewritten by the optimiser's launch
                if cached:
                    st.refresh_weight_images()           # they went stale while another shape was current
                self._optimizer.attach_weight_images(st, self._IMAGE_ROLES)
        self._step_obj, self._key = st, key
        return st, cached

    def _run(self, last_conv, labels_action, labels_pose, pose_valid):
        head = self.head
        n, c = last_conv.shape[0], last_conv.shape[-1]
        X = last_conv.detach().contiguous().view(n, -1, c)
        if self.pose_form:
            J = self.params['pose_w2'].shape[1]
            labels_pose = labels_pose.contiguous().float().view(n, -1, J)
            if pose_valid.dtype == torch.bool:
                pose_valid = pose_valid.to(torch.uint8)
            pose_valid = pose_valid.contiguous()
        if self._optimizer is not None:
            self._optimizer.check_fresh()                # weights written behind the optimiser's back?
        key = (tuple(X.shape), X.dtype, self._preact)
        st, cached = self._select(key, X, labels_action, labels_pose, pose_valid)
        if cached and self.pose_form:
            st.rebind(X=X, labels=labels_action, pose_labels=labels_pose, pose_valid=pose_valid, offset=head._step)
        elif cached:
            st.rebind(X=X, labels=labels_action, offset=head._step)
        if self.probe_events is not None:                # measurement aid (tools/bench_e2e.py): the call's device time
            self.probe_events[0].record()
        st.run()
        if self.probe_events is not None:
            self.probe_events[1].record()
        head._step += 1                                  # a fresh dropout mask per step
        if self.pose_form:                               # loss.py:70 then :75 -- tf.GraphKeys.LOSSES order
            self._losses = [st.loss_pose[0].clone(), st.loss_action[0].clone()]
            total = self._losses[0] + self._losses[1]
        else:
            self._losses = [st.loss[0].clone()]
            total = self._losses[0].clone()
        if self.loss_scale != 1.0:       # the entry points scale the GRADIENT (grad_scale); the value follows here
            total = total * self.loss_scale
        return total

    def __call__(self, images, labels_action, labels_pose=None, pose_valid=None):
        last_conv, self._preact = self.network_fn.features(images)
        if last_conv.dim() != 4:
            raise ValueError('FusedHeadStep: [N,H,W,C] feature maps (video input takes the module path)')
        if self._preact and not self.head.can_fuse_input_relu(last_conv.dtype):
            last_conv, self._preact = torch.relu(last_conv), False
        if self.pose_form and (labels_pose is None or pose_valid is None):
            raise ValueError('FusedHeadStep: the cfg 003 form needs labels_pose [N,H,W,J] and pose_valid [N,J]')
        # one autograd node either way: with a differentiated backbone it hands conv5's gradient upstream, without
        # one (head-only training) it still scales the bucket by whatever coefficient `total` is given
        total = _FusedHeadFunction.apply(last_conv, self._anchor, self, labels_action, labels_pose, pose_valid)
        st = self._step_obj
        n, h, w = last_conv.shape[:3]
        ep = {'Logits': st.logits, 'PosePrelogitsBasedAttention': st.att.view(n, h, w, -1),
              'Losses': [l.detach() for l in self._losses]}
        if self.pose_form:
            ep['PoseLogits'] = st.Pl.view(n, h, w, -1)
        return total, ep
