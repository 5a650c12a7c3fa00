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

    def __init__(self, bucket_att: torch.Tensor, bucket_td: torch.Tensor, comm_att, comm_td, device):
        from .custom_ops import custom_ops_factory as cof
        self._cof = cof
        self.bucket_att, self.bucket_td = bucket_att, bucket_td
        self.comm_att, self.comm_td = comm_att, comm_td
        self.compute = torch.cuda.current_stream(device)
        self.side = torch.cuda.Stream(device)
        self.ready, self.td_done = torch.cuda.Event(), torch.cuda.Event()
        self.ready.record(self.compute)          # materialise the handles; both start out signalled
        self.td_done.record(self.compute)
        torch.cuda.synchronize(device)
        self.hooks = cof.make_hooks(grad_ready=self.ready, td_weights_ready=self.td_done)

    def after_backward(self, update_td=None, update_att=None) -> None:
        """Call right after `attn_pool_bwd` was enqueued on the compute stream.  `update_td()` /
        `update_att()`: optional optimizer launches for the two parts; `update_td` runs with the
        communication stream current."""
        self.side.wait_event(self.ready)
        self.comm_td.all_reduce_(self.bucket_td, self.side)
        if update_td is not None:
            with torch.cuda.stream(self.side):
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


class MomentumSGD:
    """tf.train.MomentumOptimizer(lr, momentum) (src/train.py:90-94): acc = m*acc + g;
    w -= lr*acc (no Nesterov, no dampening) on the flat bucket layout, with the slim L2
    regulariser's gradient `wd * w` folded in for the names in `regularized`
    (resnet_utils.py:241: conv weights only).  On a GPU the whole update is ONE fused HIP launch
    (`apa_momentum_sgd_step`); CPU tensors (the gloo tests) take the equivalent torch expressions."""

    def __init__(self, params: Dict[str, torch.Tensor], bucket: GradientBucket, lr: float,
                 momentum: float = 0.9, weight_decay: float = 0.0, regularized: Sequence[str] = (),
                 bf16_shadows: Optional[Dict[str, torch.Tensor]] = None):
        """`bf16_shadows` {name: bf16 tensor}: operand copies the bf16 MFMA products read (the pose head's W1 for
        `cof.PoseAttnTrainStep(w1_bf16=...)`), rewritten from the updated weights by the update's own launch."""
        self.shadows = dict(bf16_shadows or {})
        self.params = params
        for name, t in self.shadows.items():
            if name not in params or t.dtype != torch.bfloat16 or t.numel() != params[name].numel():
                raise ValueError('bf16 shadow %r: a bfloat16 tensor with the element count of that parameter' % name)
        self.refresh_shadows()          # a shadow is current from the start, not only after the first update
        self.bucket = bucket
        self.lr = lr
        self.momentum = momentum
        self.acc = torch.zeros_like(bucket.flat)
        reg = set(regularized)
        self.wd = [weight_decay if n in reg else 0.0 for n in bucket.names]

    def attach_weight_images(self, step, names: Dict[str, str]) -> None:
        """Keep the per-class head's operand images current in THIS optimiser's launch: `step` is a
        cof.HeadTrainStep(..., weight_images=True) (or anything with `.weight_image_maps` and
        `.refresh_weight_images()`), `names` maps the roles 'Wa' / 'ba' / 'Wt' / 'bt' to parameter names of the
        bucket.  Several steps (micro-batch lanes, rotating buffer sets) may be attached; at most three images per
        parameter fit the fused launch, further ones are rebuilt by their step's own refresh launch after the
        update."""
        if not hasattr(self, 'images'):
            self.images, self._img_refresh = [], []
        count = {}
        for _, m in self.images:
            count[_] = count.get(_, 0) + 1
        fits = True
        new = []
        for role, m in step.weight_image_maps:
            seg = self.bucket.names.index(names[role])
            count[seg] = count.get(seg, 0) + 1
            fits = fits and count[seg] <= 3
            new.append((seg, m))
        if fits:
            self.images += new
        else:
            self._img_refresh.append(step)

    def add_image(self, name: str, image_map, refresh=None) -> None:
        """One more operand image of parameter `name` (a cof.ApaWeightImage, e.g. cof.pose_w2t_image_map: the bf16
        transposed copy of the pose head's W2) to be rewritten by this optimiser's launch; `refresh()` rebuilds it
        from the weights where the launch cannot (more than three images on one parameter; the adaptive optimisers,
        whose launches carry no image maps)."""
        if not hasattr(self, 'images'):
            self.images, self._img_refresh = [], []
        seg = self.bucket.names.index(name)
        fused = isinstance(self, MomentumSGD) and not isinstance(self, _AdaptiveOptimizer)
        if fused and sum(1 for s_, _ in self.images if s_ == seg) < 3:
            self.images.append((seg, image_map))
        elif refresh is not None:
            self._img_refresh.append(type('ImageRefresh', (), {'refresh_weight_images': staticmethod(refresh)})())
        else:
            raise ValueError('add_image(%r): this optimiser cannot rewrite the image in its launch; pass refresh=' % name)

    def refresh_shadows(self) -> None:
        """Rewrite every bf16 operand copy from its parameter: at construction, and after the weights were
        changed behind the optimiser's back (load_state_dict, a checkpoint restore)."""
        with torch.no_grad():
            for name, t in self.shadows.items():
                t.copy_(self.params[name].data.reshape(t.shape))

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        lr = self.lr if lr is None else lr
        ws = [self.params[n].data for n in self.bucket.names]
        if self.bucket.flat.is_cuda:
            from .custom_ops import custom_ops_factory as cof
            sh = [self.shadows.get(n) for n in self.bucket.names] if self.shadows else None
            cof.momentum_sgd_step(ws, self.wd, self.bucket.flat, self.acc, lr, self.momentum, grad_scale, shadows=sh,
                                  images=getattr(self, 'images', None))
            for st in getattr(self, '_img_refresh', ()):
                st.refresh_weight_images()
            return
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


class _AdaptiveOptimizer(MomentumSGD):
    """shared plumbing of Adam / RMSProp below: two slot buffers in the bucket's layout, the L2 term folded in"""

    def __init__(self, params, bucket, lr, weight_decay=0.0, regularized=(), bf16_shadows=None):
        super().__init__(params, bucket, lr, 0.0, weight_decay, regularized, bf16_shadows)
        self.slot2 = torch.zeros_like(bucket.flat)          # self.acc is the first slot

    def attach_weight_images(self, step, names=None) -> None:
        """the adaptive launches carry no image maps: an attached step rebuilds its images after each update"""
        if not hasattr(self, '_img_refresh'):
            self._img_refresh = []
        self._img_refresh.append(step)

    def _refresh_images(self) -> None:
        for st in getattr(self, '_img_refresh', ()):
            st.refresh_weight_images()

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
                 bf16_shadows=None):
        super().__init__(params, bucket, lr, weight_decay, regularized, bf16_shadows)
        self.beta1, self.beta2, self.epsilon, self.t = float(beta1), float(beta2), float(epsilon), 0

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        lr = self.lr if lr is None else lr
        self.t += 1
        if self.bucket.flat.is_cuda:
            from .custom_ops import custom_ops_factory as cof
            ws = [self.params[n].data for n in self.bucket.names]
            sh = [self.shadows.get(n) for n in self.bucket.names] if self.shadows else None
            cof.adam_step(ws, self.wd, self.bucket.flat, self.acc, self.slot2, lr, self.t, self.beta1, self.beta2,
                          self.epsilon, grad_scale, shadows=sh)
            self._refresh_images()
            return
        lr_t = lr * (1.0 - self.beta2 ** self.t) ** 0.5 / (1.0 - self.beta1 ** self.t)
        for w, g, m, v in self._segments(grad_scale):
            m.add_((g - m) * (1.0 - self.beta1))
            v.add_((g * g - v) * (1.0 - self.beta2))
            w.sub_(m * lr_t / (v.sqrt() + self.epsilon))
        self.refresh_shadows()


class RMSProp(_AdaptiveOptimizer):
    """tf.train.RMSPropOptimizer(lr, decay, momentum, epsilon) (src/train.py:95-100; TF 1.1 training/rmsprop.py, not
    centered):  ms += (g^2 - ms)(1 - decay);  mom = momentum mom + lr g / sqrt(ms + eps);  w -= mom;  `ms` starts at
    ONE.  One fused HIP launch (`apa_rmsprop_step`) on a GPU."""

    def __init__(self, params, bucket, lr, decay=0.9, momentum=0.0, epsilon=1e-10, weight_decay=0.0, regularized=(),
                 bf16_shadows=None):
        super().__init__(params, bucket, lr, weight_decay, regularized, bf16_shadows)
        self.decay, self.momentum, self.epsilon = float(decay), float(momentum), float(epsilon)
        self.acc.fill_(1.0)                                  # rmsprop.py _create_slots: init_rms = ones

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        lr = self.lr if lr is None else lr
        if self.bucket.flat.is_cuda:
            from .custom_ops import custom_ops_factory as cof
            ws = [self.params[n].data for n in self.bucket.names]
            sh = [self.shadows.get(n) for n in self.bucket.names] if self.shadows else None
            cof.rmsprop_step(ws, self.wd, self.bucket.flat, self.acc, self.slot2, lr, self.decay, self.momentum,
                             self.epsilon, grad_scale, shadows=sh)
            self._refresh_images()
            return
        for w, g, ms, mom in self._segments(grad_scale):
            ms.add_((g * g - ms) * (1.0 - self.decay))
            mom.mul_(self.momentum).add_(g * lr / (ms + self.epsilon).sqrt())
            w.sub_(mom)
        self.refresh_shadows()


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
                        regularized: Sequence[str] = (), bf16_shadows: Optional[Dict[str, torch.Tensor]] = None):
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
                    regularized=regularized, bf16_shadows=bf16_shadows)
    if kind == 'rmsprop':
        if 'RMSPROP_DECAY' not in cfg.TRAIN:
            raise AttributeError('RMSPROP_DECAY')            # src/train.py:98 on the reference's own config table
        return RMSProp(params, bucket, lr=learning_rate, decay=float(cfg.TRAIN.RMSPROP_DECAY),
                       momentum=float(cfg.TRAIN.MOMENTUM), epsilon=float(cfg.TRAIN.OPT_EPSILON), weight_decay=wd,
                       regularized=regularized, bf16_shadows=bf16_shadows)
    if kind == 'momentum':
        momentum = float(cfg.TRAIN.MOMENTUM)
    elif kind == 'sgd':
        momentum = 0.0
    else:
        raise ValueError('Optimizer [%s] was not recognized' % kind)
    return MomentumSGD(params, bucket, lr=learning_rate, momentum=momentum,
                       weight_decay=float(cfg.TRAIN.WEIGHT_DECAY), regularized=regularized, bf16_shadows=bf16_shadows)


class _FusedHeadFunction(torch.autograd.Function):
    """autograd node of FusedHeadStep: the forward IS the whole head step (forward + losses + backward, one host
    call); the backward hands the stored conv5 gradient on to the backbone."""

    @staticmethod
    def forward(ctx, last_conv, owner, labels_action, labels_pose, pose_valid):
        total = owner._run(last_conv, labels_action, labels_pose, pose_valid)
        ctx.owner = owner
        ctx.xshape = last_conv.shape
        return total

    @staticmethod
    def backward(ctx, g_total):
        o = ctx.owner
        dX = o._dX.view(ctx.xshape)
        if not o.assume_unit_upstream:        # total entered the differentiated scalar with some coefficient
            dX = dX * g_total.to(dX.dtype)
            for n in o._written:
                o.bucket.views[n].mul_(g_total)
        return dX, None, None, None, None


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
      in tf.GraphKeys.LOSSES order, detached).
    * the head's gradients are written straight into `bucket` views, in the step's own launches; the bf16 operand
      copy of the pose head's W1 is owned by the optimiser (`make_optimizer`: `bf16_shadows`) and rewritten by its
      update launch, so no conversion kernel runs in the step.
    * `assume_unit_upstream=True`: the caller promises `total` enters the differentiated scalar with coefficient 1
      (`(total + other).backward()`), which saves one pass over the [N,H,W,C] gradient; the default multiplies by
      whatever arrives."""

    def __init__(self, network_fn, cfg, loss_scale: float = 1.0, assume_unit_upstream: bool = False):
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
        self._step_obj = None
        self._key = None
        self._dX = None
        self.w1_shadow = None
        self.w2t_image = None
        self._optimizer = None
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

    def make_optimizer(self, learning_rate: float):
        """deploy.configure_optimizer on the head's parameters and this object's bucket, with the bf16 copy of the
        pose head's W1 as a shadow of the update launch when the cfg 003 step will read one (bf16 features)."""
        shadows = None
        if self.pose_form:
            w1 = self.params['pose_w1']
            self.w1_shadow = torch.empty(w1.shape, dtype=torch.bfloat16, device=w1.device)
            shadows = {'pose_w1': self.w1_shadow}
        self._step_obj = None                        # re-bind with the shadow / the weight images
        self._optimizer = configure_optimizer(self.cfg, {n: p.data for n, p in self.params.items()}, self.bucket,
                                              learning_rate, regularized=self.regularized, bf16_shadows=shadows)
        w2 = self.params['pose_w2'].data
        if self.pose_form and w2.is_cuda and w2.shape[1] <= 16:
            # the Pl product reads W2^T as a ready-made bf16 image, rewritten by the optimiser's launch as well
            from .custom_ops import custom_ops_factory as cof
            self.w2t_image = cof.pose_w2t_image(w2)
            img = self.w2t_image
            self._optimizer.add_image('pose_w2', cof.pose_w2t_image_map(img, w2),
                                      refresh=lambda: img[:w2.shape[1], :w2.shape[0]].copy_(w2.t()))
        return self._optimizer

    def _bind(self, X, labels_action, labels_pose, pose_valid):
        from .custom_ops import custom_ops_factory as cof
        head, tr, v = self.head, self.cfg.TRAIN, self.bucket.views
        flags = cof.attn_flags(head.softmax_att, head.relu_att, True, self._preact)
        self._dX = torch.empty_like(X)
        p = {n: t.data for n, t in self.params.items()}
        if self.pose_form:
            shadow = self.w1_shadow if X.dtype == torch.bfloat16 else None
            return cof.PoseAttnTrainStep(
                X, (p['pose_w1'], p['pose_b1'], p['pose_w2'], p['pose_b2'], p['att_weights'], p['att_biases'],
                    p['td_weights'], p['td_biases']), labels_action, labels_pose, pose_valid,
                (self._dX, v['pose_w1'], v['pose_b1'], v['pose_w2'], v['pose_b2'], v['att_weights'], v['att_biases'],
                 v['td_weights'], v['td_biases']), flags=flags, keep_prob=head.keep_prob, seed=head.seed,
                offset=head._step, action_wt=float(tr.LOSS_FN_ACTION_WT), pose_wt=float(tr.LOSS_FN_POSE_WT),
                grad_scale=self.loss_scale, w1_bf16=shadow,
                w2t_bf16=self.w2t_image if X.dtype == torch.bfloat16 else None)
        st = cof.HeadTrainStep(
            X, X, p['att_weights'], p['att_biases'], p['td_weights'], p['td_biases'], labels_action,
            (self._dX, None, v['att_weights'], v['att_biases'], v['td_weights'], v['td_biases']), flags=flags,
            keep_prob=head.keep_prob, seed=head.seed, offset=head._step, loss_wt=float(tr.LOSS_FN_ACTION_WT),
            grad_scale=self.loss_scale, hooks=head.hooks,
            weight_images=bool(head.per_class and self._optimizer is not None))
        if st.weight_image_maps:      # per-class maps: the operand images are rewritten by the optimiser's launch
            self._optimizer.attach_weight_images(st, {'Wa': 'att_weights', 'ba': 'att_biases', 'Wt': 'td_weights',
                                                      'bt': 'td_biases'})
        return st

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
        key = (tuple(X.shape), X.dtype, self._preact)
        if self._step_obj is None or key != self._key:
            self._step_obj, self._key = self._bind(X, labels_action, labels_pose, pose_valid), key
        elif self.pose_form:
            self._step_obj.rebind(X=X, labels=labels_action, pose_labels=labels_pose, pose_valid=pose_valid,
                                  offset=head._step)
        else:
            self._step_obj.rebind(X=X, labels=labels_action, offset=head._step)
        st = self._step_obj
        if self.probe_events is not None:                # measurement aid (tools/bench_e2e.py): the call's device time
            self.probe_events[0].record()
        st.run()
        if self.probe_events is not None:
            self.probe_events[1].record()
        head._step += 1                                  # a fresh dropout mask per step
        if self.pose_form:                               # loss.py:70 then :75 -- tf.GraphKeys.LOSSES order
            self._losses = [st.loss_pose[0], st.loss_action[0]]
            total = st.loss_pose[0] + st.loss_action[0]
        else:
            self._losses = [st.loss[0]]
            total = st.loss[0].clone()
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
        if last_conv.requires_grad:
            total = _FusedHeadFunction.apply(last_conv, self, labels_action, labels_pose, pose_valid)
        else:       # nothing upstream to differentiate (head-only training): `total.backward()` stays legal, a no-op
            total = self._run(last_conv, labels_action, labels_pose, pose_valid).requires_grad_(True)
        st = self._step_obj
        n, h, w = last_conv.shape[:3]
        ep = {'Logits': st.logits, 'PosePrelogitsBasedAttention': st.att.view(n, h, w, -1),
              'Losses': [l.detach() for l in self._losses]}
        if self.pose_form:
            ep['PoseLogits'] = st.Pl.view(n, h, w, -1)
        return total, ep
