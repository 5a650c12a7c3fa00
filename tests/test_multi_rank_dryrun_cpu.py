"""Dry run of the N > 1 schedule on TWO CPU ranks.

No round had a node with more than one GPU: RCCL has only ever seen one-rank groups, so the two-communicator
overlapped gradient sum (deploy.OverlappedGradientSum + rccl.RcclCommunicator, bench.py --gpus N) ran its
bookkeeping only where it cannot go wrong.  Here the PRODUCT classes run on two gloo ranks against stand-ins for the
two things a CPU lacks:

  * `FakeNccl`   -- a library object with librccl's entry points (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
                    ncclCommCount, ncclCommUserRank, ncclCommDestroy); ncclAllReduce is a gloo all-reduce on the
                    memory the pointer argument names, and a collective on a communicator that is still "in flight"
                    on another stream is an error;
  * `FakeRuntime`-- streams and events that execute eagerly and keep a log (deploy.HipRuntime's interface).

What is checked: communicator creation order and the unique-id exchange (same id on both ranks per communicator,
different ids for the two communicators, ncclCommCount == world), one communicator <-> one stream, the two event
hand-overs of every step in the order the kernels rely on, identical collective sequences on both ranks, sums equal to
the sequential `sum_clone_gradients` loop over four SGD steps, and that `--overlap auto`'s probe takes the SAME
decision on every rank even when their local clocks disagree.  What stays untested until a node exists: RCCL itself
(ring construction over xGMI, two live communicators per process), real stream concurrency and the timing."""
import ctypes
import os
import socket
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from attentionalpoolingaction_amd import deploy, rccl


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeStream:
    def __init__(self, rt, name):
        self.rt, self.name = rt, name
        self.cuda_stream = id(self) & 0x7fffffff

    def wait_event(self, ev):
        assert ev.recorded_on is not None, 'waiting for an event nobody recorded'
        self.rt.log.append(('wait', self.name, ev.name, ev.generation))

    def synchronize(self):
        self.rt.log.append(('sync', self.name))


class FakeEvent:
    def __init__(self, rt, name):
        self.rt, self.name, self.recorded_on, self.generation = rt, name, None, 0

    def record(self, stream):
        self.recorded_on = stream.name
        self.generation += 1
        self.rt.log.append(('record', stream.name, self.name, self.generation))


class FakeRuntime:
    """deploy.HipRuntime's interface with eager execution and a log"""

    def __init__(self):
        self.log = []
        self._compute = FakeStream(self, 'compute')
        self._current = self._compute
        self._n_ev = 0
        self.hooks_made = None

    def current_stream(self):
        return self._current

    def new_stream(self):
        return FakeStream(self, 'side')

    def new_event(self):
        self._n_ev += 1
        return FakeEvent(self, 'ev%d' % self._n_ev)

    def synchronize(self):
        self.log.append(('device_sync',))

    def stream(self, s):
        rt = self

        class _Ctx:
            def __enter__(self_inner):
                self_inner.prev, rt._current = rt._current, s

            def __exit__(self_inner, *a):
                rt._current = self_inner.prev
        return _Ctx()

    def make_hooks(self, grad_ready, td_weights_ready):
        self.hooks_made = (grad_ready, td_weights_ready)
        return {'grad_ready_event': grad_ready, 'td_weights_ready_event': td_weights_ready}


class FakeNccl:
    """librccl's entry points as rccl.RcclCommunicator calls them; collectives go over the gloo group"""

    def __init__(self, rt):
        self.rt = rt
        self.comms = {}          # handle -> dict(uid, rank, world, stream, calls)
        self._next = 1
        self.uids_issued = 0

    def ncclGetErrorString(self, rc):
        return b'fake nccl error %d' % rc

    def ncclGetUniqueId(self, ref):
        self.uids_issued += 1
        raw = bytes([(17 * self.uids_issued + i) & 0xff for i in range(rccl.NCCL_UNIQUE_ID_BYTES)])
        ctypes.memmove(ctypes.addressof(ref._obj), raw, len(raw))
        return 0

    def ncclCommInitRank(self, ref, world, uid, rank):
        h = self._next
        self._next += 1
        self.comms[h] = {'uid': bytes(uid.internal), 'rank': rank, 'world': world, 'stream': None, 'calls': []}
        ref._obj.value = h
        return 0

    def ncclAllReduce(self, src, dst, count, dtype, op, comm, stream):
        c = self.comms[comm.value]
        if src.value != dst.value or op != 0 or dtype != 7:
            return 5
        if c['stream'] is None:
            c['stream'] = stream.value
        elif c['stream'] != stream.value:
            return 4                     # one communicator must live on ONE stream (two collectives of one
        c['calls'].append(int(count))    # communicator must never be in flight together)
        buf = (ctypes.c_float * count).from_address(src.value)
        t = torch.frombuffer(buf, dtype=torch.float32)
        dist.all_reduce(t)               # in place on the caller's memory
        self.rt.log.append(('allreduce', comm.value, int(count)))
        return 0

    def ncclCommCount(self, comm, ref):
        ref._obj.value = self.comms[comm.value]['world']
        return 0

    def ncclCommUserRank(self, comm, ref):
        ref._obj.value = self.comms[comm.value]['rank']
        return 0

    def ncclCommDestroy(self, comm):
        self.comms.pop(comm.value, None)
        return 0


def _all_equal_across_ranks(obj, world):
    out = [None] * world
    dist.all_gather_object(out, obj)
    return all(o == out[0] for o in out), out


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rt = FakeRuntime()
        lib = FakeNccl(rt)
        # bench.py's order: the in-stream communicator first, the communication stream's second -- on EVERY rank
        # (each constructor broadcasts rank 0's id; a rank that built them the other way round would pair its
        # first communicator with the peers' second)
        comm_att = rccl.RcclCommunicator(rank, world, 'cpu', group=None, lib=lib, current_stream=rt.current_stream)
        comm_td = rccl.RcclCommunicator(rank, world, 'cpu', group=None, lib=lib, current_stream=rt.current_stream)
        assert comm_att.count() == world and comm_td.count() == world
        assert comm_att.user_rank() == rank and comm_td.user_rank() == rank
        uids = [lib.comms[c.comm.value]['uid'] for c in (comm_att, comm_td)]
        same, gathered = _all_equal_across_ranks(uids, world)
        assert same, 'the ranks disagree about a communicator\'s unique id'
        assert uids[0] != uids[1] and any(uids[0]) and any(uids[1])
        assert lib.uids_issued == (2 if rank == 0 else 0)          # only rank 0 asks the library for ids

        # a toy head: att part (8 KB in the product: here 5 floats) | td part
        g = torch.Generator().manual_seed(3)
        w0 = torch.randn(5 + 12, generator=g)
        w_seq, w_ovl = w0.clone(), w0.clone()
        bucket_seq = deploy.GradientBucket({'att': (5,), 'td': (12,)}, 'cpu')
        bucket = deploy.GradientBucket({'att': (5,), 'td': (12,)}, 'cpu')
        ogs = deploy.OverlappedGradientSum(bucket.flat[:5], bucket.flat[5:], comm_att, comm_td, 'cpu', runtime=rt)
        assert rt.hooks_made == (ogs.ready, ogs.td_done) and ogs.hooks['grad_ready_event'] is ogs.ready
        cfg = deploy.DeploymentConfig()
        lr = 0.1
        for step in range(4):
            gr = torch.Generator().manual_seed(100 * step + rank)      # every rank its own tower gradient
            grad = torch.randn(17, generator=gr) * cfg.clone_loss_scale
            # yardstick: one bucket, torch.distributed, then the update
            bucket_seq.flat.copy_(grad)
            deploy.sum_clone_gradients(bucket_seq, cfg)
            w_seq -= lr * bucket_seq.flat
            # the overlapped schedule on the fake runtime.  The library records `ready` after the first backward
            # kernel (cof hooks); played here by hand:
            bucket.flat.copy_(grad)
            mark = len(rt.log)
            ogs.ready.record(ogs.compute)
            on_stream = []

            def update_td():
                on_stream.append(rt.current_stream().name)
                w_ovl[5:] -= lr * bucket.flat[5:]

            def update_att():
                on_stream.append(rt.current_stream().name)
                w_ovl[:5] -= lr * bucket.flat[:5]
            ogs.after_backward(update_td=update_td, update_att=update_att)
            assert on_stream == ['side', 'compute']
            ev = rt.log[mark:]
            kinds = [(e[0], e[1]) + ((e[2],) if e[0] != 'allreduce' else ()) for e in ev]
            h_td, h_att = comm_td.comm.value, comm_att.comm.value
            assert kinds == [('record', 'compute', ogs.ready.name),       # grad_ready after the head kernel
                             ('wait', 'side', ogs.ready.name),            # the comm stream starts behind it
                             ('allreduce', h_td),                         # td part on its own communicator
                             ('record', 'side', ogs.td_done.name),        # what the NEXT forward's logits wait for
                             ('allreduce', h_att)], kinds                 # the 8 KB part stays in-stream
            assert ev[1][3] == ev[0][3]                                   # ... and it is THIS step's recording
            assert torch.equal(bucket.flat, bucket_seq.flat)
        assert torch.equal(w_ovl, w_seq)
        # one communicator <-> one stream, identical collective sequences on every rank
        assert lib.comms[h_td]['stream'] == ogs.side.cuda_stream and lib.comms[h_att]['stream'] == ogs.compute.cuda_stream
        same, _ = _all_equal_across_ranks([lib.comms[h_att]['calls'], lib.comms[h_td]['calls']], world)
        assert same and lib.comms[h_att]['calls'] == [5] * 4 and lib.comms[h_td]['calls'] == [12] * 4
        # a collective of the td communicator enqueued on the compute stream is refused by the stand-in
        # (RCCL would not refuse it -- it would deadlock or corrupt; the schedule must never do it)
        try:
            comm_td.all_reduce_(bucket.flat[5:], ogs.compute)
            raise AssertionError('expected the stand-in to refuse a second stream on one communicator')
        except RuntimeError as e:
            assert 'ncclAllReduce failed' in str(e)

        # --overlap auto: the ranks' local clocks disagree (rank 1 finds the two-stream schedule slow, rank 0 the
        # in-stream one) -- the MAX-reduced medians are the same on both, so is the decision
        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        slow_two = 0.004 if rank == 1 else 0.0
        slow_one = 0.002 if rank == 0 else 0.0
        keep, us_two, us_one = deploy.probe_overlap_schedule(
            lambda: time.sleep(slow_two), lambda: time.sleep(slow_one), dist.barrier, reduce_max, loops=3, steps=2)
        same, got = _all_equal_across_ranks((keep, round(us_two, 6), round(us_one, 6)), world)
        assert same, got
        assert keep is False and us_two > us_one > 1000.0        # rank 1's 4 ms dominate rank 0's 2 ms
        ogs.close()
        comm_att.close()
        comm_td.close()
        assert not lib.comms
        np.save(os.path.join(out_dir, 'w_rank%d.npy' % rank), w_ovl.numpy())
    finally:
        dist.destroy_process_group()


def test_overlapped_two_communicator_schedule_on_two_gloo_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w0, w1 = np.load(tmp_path / 'w_rank0.npy'), np.load(tmp_path / 'w_rank1.npy')
    np.testing.assert_array_equal(w0, w1)                 # replicas stay identical


def test_bench_refuses_a_communicator_that_does_not_span_the_ranks():
    """bench.py --gpus N: `ncclCommCount != WORLD_SIZE` ends the run with a message, not with a scaling line"""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')).read()
    assert 'comm.count() != max(world, 1)' in src and 'ncclCommCount must equal WORLD_SIZE' in src
