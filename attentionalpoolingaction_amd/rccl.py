"""Direct RCCL binding (ctypes) for the data-parallel gradient sum.

`models/slim/deployment/model_deploy.py:421-451` sums the per-tower gradients with `tf.add_n` on the
parameter device; here every GPU is its own process and the sum is ONE in-stream
`ncclAllReduce` of the flat gradient bucket over xGMI.  torch.distributed's NCCL backend is RCCL
too, but each `dist.all_reduce` call costs ~20 us of host time and bounces through a private
stream (measured: a 55 us head step becomes 78 us on a ONE-rank group).  Calling librccl directly
enqueues the collective on the compute stream itself: no stream hop, ~3 us of host time, and the
launch is hipGraph-capturable.

Bootstrap: rank 0 creates the ncclUniqueId, any existing torch.distributed group (gloo is enough)
broadcasts its 128 bytes, every rank calls ncclCommInitRank.
"""
import ctypes
import os
from typing import Optional

import torch

NCCL_UNIQUE_ID_BYTES = 128
_DTYPES = {torch.float32: 7, torch.float64: 8, torch.float16: 6, torch.bfloat16: 9,
           torch.int32: 2, torch.int64: 4, torch.uint8: 1, torch.int8: 0}
_NCCL_SUM = 0
_lib = None


class _UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)]   # (c_char would truncate at a NUL)


def load_rccl() -> ctypes.CDLL:
    """The librccl.so PyTorch-ROCm itself links (torch/lib), else the system one."""
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'),
             '/opt/rocm/lib/librccl.so', 'librccl.so', 'librccl.so.1']
    err = None
    for c in cands:
        try:
            lib = ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError as e:          # keep looking
            err = e
    else:
        raise OSError('librccl.so not found ({})'.format(err))
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclCommUserRank.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    _lib = lib
    return lib


def _check(lib, rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError('{} failed: {}'.format(what, lib.ncclGetErrorString(rc).decode()))


def exchange_unique_id(raw: bytes, rank: int, world_size: int, group=None, device=None) -> bytes:
    """Rank 0's 128-byte ncclUniqueId to every rank through a torch.distributed group (gloo: CPU
    tensor; nccl: device tensor).  Split out so the bootstrap is testable without RCCL."""
    if world_size == 1:
        return raw
    import torch.distributed as dist
    buf = torch.frombuffer(bytearray(raw if rank == 0 else bytes(NCCL_UNIQUE_ID_BYTES)),
                           dtype=torch.uint8).clone()
    if dist.get_backend(group) == 'nccl':
        buf = buf.to(device)
    dist.broadcast(buf, src=0, group=group)
    return bytes(buf.cpu().numpy().tobytes())


class RcclCommunicator:
    """One communicator per process (= per GPU).  `group`: an initialised torch.distributed group
    used only to ship the unique id (None when world_size == 1).

    `lib` / `current_stream` (round 6): the library object and the stream source are injectable so that the CALL
    SEQUENCE of this class -- unique id on rank 0, the 128-byte broadcast, ncclCommInitRank on every rank, one
    ncclAllReduce per call on the given stream -- can be driven on two CPU ranks against a stand-in whose
    ncclAllReduce is a gloo all-reduce (tests/test_multi_rank_dryrun_cpu.py).  No GPU node with more than one rank
    was available to any round; the product always runs with the defaults (librccl, torch's current HIP stream)."""

    def __init__(self, rank: int, world_size: int, device: torch.device, group=None, lib=None, current_stream=None):
        self._stub = lib is not None
        self.lib = load_rccl() if lib is None else lib
        self._current_stream = current_stream
        self.rank, self.world_size, self.device = rank, world_size, torch.device(device)
        uid = _UniqueId()
        if rank == 0:
            _check(self.lib, self.lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        raw = exchange_unique_id(ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES), rank,
                                 world_size, group, self.device)
        assert len(raw) == NCCL_UNIQUE_ID_BYTES
        ctypes.memmove(ctypes.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self.comm = ctypes.c_void_p()
        if self._stub:
            _check(self.lib, self.lib.ncclCommInitRank(ctypes.byref(self.comm), world_size, uid, rank),
                   'ncclCommInitRank')
        else:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.ncclCommInitRank(ctypes.byref(self.comm), world_size, uid, rank),
                       'ncclCommInitRank')

    def all_reduce_(self, t: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """In-place SUM over all ranks, enqueued on `stream` (default: the current stream)."""
        if not (t.is_cuda or self._stub) or not t.is_contiguous():
            raise ValueError('all_reduce_ needs a contiguous device tensor')
        if stream is None:
            stream = self._current_stream() if self._current_stream is not None else \
                torch.cuda.current_stream(self.device)
        st = stream.cuda_stream
        _check(self.lib, self.lib.ncclAllReduce(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()),
                                                t.numel(), _DTYPES[t.dtype], _NCCL_SUM, self.comm,
                                                ctypes.c_void_p(st)), 'ncclAllReduce')
        return t

    def count(self) -> int:
        """ncclCommCount: the number of ranks RCCL itself sees in this communicator."""
        n = ctypes.c_int(-1)
        _check(self.lib, self.lib.ncclCommCount(self.comm, ctypes.byref(n)), 'ncclCommCount')
        return int(n.value)

    def user_rank(self) -> int:
        r = ctypes.c_int(-1)
        _check(self.lib, self.lib.ncclCommUserRank(self.comm, ctypes.byref(r)), 'ncclCommUserRank')
        return int(r.value)

    def measure_all_reduce_us(self, numel: int, iters: int = 20, warm: int = 5) -> float:
        """Average device time of one in-stream all-reduce of `numel` fp32 elements (back-to-back launches between
        two events on the current stream): the figure a trainer weighs against the cost of hiding it."""
        buf = torch.zeros(numel, dtype=torch.float32, device=self.device)
        for _ in range(warm):
            self.all_reduce_(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(self.device)
        e0.record()
        for _ in range(iters):
            self.all_reduce_(buf)
        e1.record()
        torch.cuda.synchronize(self.device)
        return e0.elapsed_time(e1) * 1e3 / iters

    def close(self) -> None:
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()
