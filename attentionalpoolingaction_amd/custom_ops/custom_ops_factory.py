"""Loader + thin python wrappers for libapa_hip.so (the MI355X attentional-pooling library).

Shaped like the reference's plugin layer, /root/reference/src/custom_ops/custom_ops_factory.py:
the shared object lives next to this file (reference :7-18 resolves `<op>.so` relative to
`__file__` and calls tf.load_op_library at import time) and every op is exposed as a small python
wrapper.  Here the library is a plain C-ABI .so (include/apa.h) bound with ctypes; tensors are
torch tensors used purely as device-memory handles (data_ptr + current stream).

There is NO CPU fallback: if the library is missing, or a tensor is not on the GPU, the wrappers
raise.  The oracle under /oracle is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint, c_uint8, c_uint64,
                    c_void_p)
from typing import Optional, Tuple

import numpy as np
import torch  # imported BEFORE the CDLL so that libamdhip64.so.7 resolves to torch's runtime

cur_path = os.path.realpath(__file__)
ROOT_PATH = os.path.dirname(cur_path)
LIB_NAME = 'libapa_hip.so'
LIB_PATH = os.path.join(ROOT_PATH, LIB_NAME)

APA_DTYPE_F32 = 0
APA_DTYPE_BF16 = 1
APA_FLAG_SOFTMAX_ATT = 1
APA_FLAG_RELU_ATT = 2
APA_FLAG_TRAIN = 4
APA_FLAG_RNG_DEVICE = 8
APA_FLAG_RELU_INPUT = 16  # X in memory is the pre-activation map: relu fused into both passes
APA_FLAG_DXATT_RANK1 = 32  # attn_pool_bwd returns dZ [N*P] instead of dXatt = dZ (x) Wa
APA_FLAG_RNG_EXTERNAL = 128  # `seed` is the address of a caller-supplied, bit-packed dropout keep mask
APA_FLAG_WEIGHT_IMAGES = 256  # per-class maps: the operand images in the workspace are kept current by the caller
APA_WIMG_MAX = 12
APA_WIMG_ROLES = {0: 'Wa', 1: 'ba', 2: 'Wt', 3: 'bt'}

# every symbol include/apa.h declares: name -> (restype, argtypes)
_SIGNATURES = {
    'apa_version': (c_int, []),
    'apa_last_error': (c_char_p, []),
    'apa_status_string': (c_char_p, [c_int]),
    'apa_attn_pool_workspace_bytes': (c_size_t, [c_int] * 6 + [c_uint]),
    'apa_attn_pool_fwd': (c_int, [c_void_p] * 12 + [c_size_t] + [c_int] * 6 +
                          [c_uint, c_float, c_uint64, c_uint64, c_int, c_void_p]),
    'apa_attn_pool_bwd': (c_int, [c_void_p] * 17 + [c_size_t] + [c_int] * 6 +
                          [c_uint, c_float, c_uint64, c_uint64, c_int, c_void_p]),
    # the *_ex forms take a `const apa_hooks*` first (NULL = no hooks)
    'apa_attn_pool_fwd_ex': (c_int, [c_void_p] * 13 + [c_size_t] + [c_int] * 6 +
                             [c_uint, c_float, c_uint64, c_uint64, c_int, c_void_p]),
    'apa_attn_pool_bwd_ex': (c_int, [c_void_p] * 18 + [c_size_t] + [c_int] * 6 +
                             [c_uint, c_float, c_uint64, c_uint64, c_int, c_void_p]),
    'apa_dropout_mask': (c_int, [c_void_p, c_size_t, c_float, c_uint64, c_uint64, c_void_p]),
    'apa_pose_head_workspace_bytes': (c_size_t, [c_int] * 6),
    'apa_pose_head_fwd': (c_int, [c_void_p] * 8 + [c_size_t] + [c_int] * 6 + [c_void_p]),
    'apa_pose_head_bwd': (c_int, [c_void_p] * 7 + [c_int] + [c_void_p] * 5 + [c_size_t] + [c_int] * 6 +
                          [c_void_p]),
    'apa_pose_head_bwd_rank1ext': (c_int, [c_void_p] * 8 + [c_int] + [c_void_p] * 5 + [c_size_t] + [c_int] * 6 +
                                   [c_void_p]),
    'apa_softmax_xent_fwd_bwd': (c_int, [c_void_p] * 6 + [c_int, c_int, c_float, c_float, c_void_p]),
    'apa_pose_l2_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'apa_pose_l2_loss_fwd_bwd': (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_int, c_int, c_float,
                                                          c_float, c_void_p]),
    'apa_action_loss_fwd_bwd': (c_int, [c_int] + [c_void_p] * 4 + [c_int, c_int, c_float, c_float, c_float, c_void_p]),
    'apa_pose_sampled_loss_fwd_bwd': (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_float, c_void_p]),
    'apa_resize_bilinear_tf1': (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'apa_pose_to_heatmap_out_ht': (c_int64, [c_int64, c_int64, c_int64]),
    'apa_pose_to_heatmap': (c_int, [POINTER(c_int64), c_int64, c_int64, c_int64, c_int64, c_int,
                                    c_float, c_int, POINTER(c_float), POINTER(c_uint8)]),
    'apa_spatial_mean_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'apa_zero_out_channels': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'apa_pose_label_replay_resize': (c_int, [POINTER(c_uint8)] + [c_int] * 11 + [c_float, POINTER(c_float)]),
    'apa_pose_labels_device': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    'apa_frame_pool_fwd': (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    'apa_frame_pool_bwd': (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p]),
    'apa_attn_head_train_step': (c_int, [c_void_p] * 7 + [c_float, c_float] + [c_void_p] * 13 +
                                 [c_size_t] + [c_int] * 6 + [c_uint, c_float, c_uint64, c_uint64, c_int,
                                                             c_void_p]),
    # ..._WITH_POSE_FEAT: `const apa_concat_feat*`, `const apa_hooks*`, then the plain arguments
    'apa_attn_pool_fwd_cat': (c_int, [c_void_p] * 14 + [c_size_t] + [c_int] * 6 +
                              [c_uint, c_float, c_uint64, c_uint64, c_int, c_void_p]),
    'apa_attn_pool_bwd_cat': (c_int, [c_void_p] * 19 + [c_size_t] + [c_int] * 6 +
                              [c_uint, c_float, c_uint64, c_uint64, c_int, c_void_p]),
    'apa_attn_head_train_step_ex': (c_int, [c_void_p] * 8 + [c_float, c_float] + [c_void_p] * 13 +
                                    [c_size_t] + [c_int] * 6 + [c_uint, c_float, c_uint64, c_uint64, c_int,
                                                                c_void_p]),
    'apa_attn_head_eval_step': (c_int, [c_void_p] * 15 + [c_size_t] + [c_int] * 6 + [c_uint, c_int, c_void_p]),
    'apa_momentum_sgd_step': (c_int, [c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_float),
                                      c_void_p, c_void_p, c_float, c_float, c_float, c_void_p]),
    'apa_momentum_sgd_step_shadow': (c_int, [c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_float),
                                             c_void_p, c_void_p, c_float, c_float, c_float, POINTER(c_void_p),
                                             c_void_p]),
    'apa_per_class_weight_images': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 6 + [c_void_p, POINTER(c_int),
                                                                                       c_void_p]),
    'apa_momentum_sgd_step_images': (c_int, [c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_float),
                                             c_void_p, c_void_p, c_float, c_float, c_float, POINTER(c_void_p),
                                             c_void_p, POINTER(c_int), c_int, c_void_p]),
    'apa_adam_step': (c_int, [c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_float), c_void_p, c_void_p,
                              c_void_p, c_float, c_float, c_float, c_float, c_float, POINTER(c_void_p), c_void_p]),
    'apa_rmsprop_step': (c_int, [c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_float), c_void_p, c_void_p,
                                 c_void_p, c_float, c_float, c_float, c_float, c_float, POINTER(c_void_p), c_void_p]),
    'apa_pose_attn_train_step': (c_int, [c_void_p] + [c_int] * 6 + [c_uint, c_float, ctypes.c_uint64,
                                                                    ctypes.c_uint64, c_int, c_void_p]),
    'apa_accumulate_gradients': (c_int, [c_void_p, POINTER(c_void_p), c_int, c_size_t, c_float, c_void_p]),
    'apa_accumulate_gradients_div': (c_int, [c_void_p, POINTER(c_void_p), c_int, c_size_t, c_float, c_void_p]),
    'apa_prof_event_create': (c_int, [POINTER(c_void_p)]),
    'apa_prof_event_destroy': (c_int, [c_void_p]),
    'apa_prof_event_record': (c_int, [c_void_p, c_void_p]),
    'apa_prof_event_elapsed_ms': (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
}

_lib = None


class ApaError(RuntimeError):
    pass


def exported_symbols():
    """Names the header declares (used by the CPU-side ABI test)."""
    return sorted(_SIGNATURES)


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen libapa_hip.so from this directory (like custom_ops_factory.py:11-18) and bind every
    entry point.  Fails loudly when the library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get('APA_LIB_PATH') or LIB_PATH      # APA_LIB_PATH: A/B experiments (tools/)
    if not os.path.exists(p):
        raise ApaError(
            '{} not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C attentionalpoolingaction_amd/csrc`. There is no CPU fallback.'.format(p))
    lib = ctypes.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == missing export
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load_library()
        raise ApaError('{} failed: {} ({})'.format(
            what, lib.apa_status_string(rc).decode(), lib.apa_last_error().decode()))


def _dev_ptr(t: Optional[torch.Tensor], name: str, dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise ApaError('{} must live in GPU memory (got device {}); the HIP path has no CPU '
                       'fallback'.format(name, t.device))
    if not t.is_contiguous():
        raise ApaError('{} must be contiguous'.format(name))
    if dtype is not None and t.dtype != dtype:
        raise ApaError('{} must be {} (got {})'.format(name, dtype, t.dtype))
    return t.data_ptr()


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _feat_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return APA_DTYPE_F32
    if t.dtype == torch.bfloat16:
        return APA_DTYPE_BF16
    raise ApaError('feature maps must be float32 or bfloat16, got {}'.format(t.dtype))


class KeepMask(object):
    """A dropout keep mask drawn OUTSIDE the library (the reference's tf.nn.dropout: floor(keep_prob + U),
    nets_factory.py:143-146,296), bit-packed for APA_FLAG_RNG_EXTERNAL: pass it as `seed=` to any of the
    attentional-pooling wrappers (forward AND backward).  Build one with pack_keep_mask()."""

    def __init__(self, bits: torch.Tensor, n_elems: int):
        self.bits, self.n_elems = bits, int(n_elems)


def pack_keep_mask(*parts: torch.Tensor, device='cuda') -> KeepMask:
    """{0,1} / bool tensors in the flat element order of the [N,P,C] feature map (for the *_cat ops followed
    by the [N,P,J] extra channels): concatenated and packed LSB-first -- bit (e & 7) of byte e >> 3 is
    element e (include/apa.h, APA_FLAG_RNG_EXTERNAL)."""
    flat = torch.cat([p.reshape(-1).to(device=device, dtype=torch.uint8) for p in parts])
    n = flat.numel()
    pad = (-n) % 64                                          # whole 8-byte words
    if pad:
        flat = torch.cat([flat, torch.zeros(pad, dtype=torch.uint8, device=flat.device)])
    w = (1 << torch.arange(8, device=flat.device, dtype=torch.int32))
    bits = ((flat.view(-1, 8) != 0).to(torch.int32) * w).sum(dim=1).to(torch.uint8).contiguous()
    return KeepMask(bits, n)


def _rng_key(seed, offset, flags):
    """-> (seed, offset, flags) as the C ABI takes them.  offset: an int is passed by value; a 1-element
    int64 CUDA tensor by address (APA_FLAG_RNG_DEVICE) so that hipGraph replays read -- and the backward
    advances -- it.  seed: an int keys the library's own counter hash; a KeepMask replays an external mask
    (APA_FLAG_RNG_EXTERNAL: its address travels in `seed`)."""
    if isinstance(seed, KeepMask):
        if isinstance(offset, torch.Tensor):
            raise ApaError('an external keep mask excludes the device-side dropout counter')
        if not seed.bits.is_cuda:
            raise ApaError('the packed keep mask must live on the GPU')
        return seed.bits.data_ptr(), 0, flags | APA_FLAG_RNG_EXTERNAL
    if isinstance(offset, torch.Tensor):
        if not (offset.is_cuda and offset.dtype == torch.int64 and offset.numel() == 1):
            raise ApaError('a device-side dropout counter must be a 1-element int64 CUDA tensor')
        return int(seed), offset.data_ptr(), flags | APA_FLAG_RNG_DEVICE
    return int(seed), int(offset), flags


class ApaHooks(ctypes.Structure):
    """`apa_hooks` of include/apa.h: per-call ordering / timing aids, passed explicitly to the *_ex
    entry points (the library keeps no per-thread or global state)."""
    _fields_ = [('grad_ready_event', c_void_p), ('td_weights_ready_event', c_void_p),
                ('prof_fwd_start', c_void_p), ('prof_fwd_stop', c_void_p),
                ('prof_bwd_start', c_void_p), ('prof_bwd_stop', c_void_p)]


def _event_handle(event):
    """torch.cuda.Event (recorded at least once, so that its handle exists), raw hipEvent_t or None."""
    if event is None:
        return None
    if isinstance(event, torch.cuda.Event):
        return event.cuda_event
    if isinstance(event, c_void_p):
        return event.value
    return int(event)


def make_hooks(grad_ready=None, td_weights_ready=None, prof_fwd=None, prof_bwd=None) -> ApaHooks:
    """Build an apa_hooks struct.  grad_ready / td_weights_ready: events (see include/apa.h);
    prof_fwd / prof_bwd: (start, stop) event pairs that receive the dispatch timestamps of the two
    streaming kernels.  Keep the returned object (and the events) alive while calls that use it are in
    flight."""
    h = ApaHooks()
    h.grad_ready_event = _event_handle(grad_ready)
    h.td_weights_ready_event = _event_handle(td_weights_ready)
    if prof_fwd is not None:
        h.prof_fwd_start, h.prof_fwd_stop = _event_handle(prof_fwd[0]), _event_handle(prof_fwd[1])
    if prof_bwd is not None:
        h.prof_bwd_start, h.prof_bwd_stop = _event_handle(prof_bwd[0]), _event_handle(prof_bwd[1])
    return h


def _hooks_ptr(hooks):
    return None if hooks is None else ctypes.addressof(hooks)


def attn_flags(softmax_att=False, relu_att=False, is_training=False, relu_input=False) -> int:
    return ((APA_FLAG_SOFTMAX_ATT if softmax_att else 0) | (APA_FLAG_RELU_ATT if relu_att else 0) |
            (APA_FLAG_TRAIN if is_training else 0) | (APA_FLAG_RELU_INPUT if relu_input else 0))


# --------------------------------------------------------------------------------------------
# attentional pooling
# --------------------------------------------------------------------------------------------
def attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags=0) -> int:
    return int(load_library().apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))


def attn_pool_fwd(X, Xatt, Wa, ba, Wt, bt, *, flags=0, keep_prob=1.0, seed=0, offset=0,
                  workspace=None, want_topdown=False, hooks=None):
    """logits, att, zsave, abar, topdown, workspace = attn_pool_fwd(...)

    X [N,P,C] (or [N,H,W,C]) f32/bf16; Xatt same tensor object as X (cfg 002) or [N,P,Ca];
    Wa [Ca,M], ba [M], Wt [C,K], bt [K] f32.  See include/apa.h: apa_attn_pool_fwd.
    """
    lib = load_library()
    N = X.shape[0]
    C = X.shape[-1]
    P = X.numel() // (N * C)
    Ca = Xatt.shape[-1]
    M = Wa.shape[1]
    K = Wt.shape[1]
    dt = _feat_dtype(X)
    dev = X.device
    logits = torch.empty((N, K), dtype=torch.float32, device=dev)
    att = torch.empty((N, P, M), dtype=torch.float32, device=dev)
    # saved for backward: M == 1 -> z [N,C] + abar [N]; per-class -> the fp32 top-down map [N,P,K]
    zsave = torch.empty((N, C) if M == 1 else (N, P, K), dtype=torch.float32, device=dev)
    abar = torch.empty((N,), dtype=torch.float32, device=dev) if M == 1 else None
    topdown = torch.empty((N, P, K), dtype=X.dtype, device=dev) if want_topdown else None
    need = int(lib.apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
    xatt_ptr = _dev_ptr(X, 'X') if Xatt is X else _dev_ptr(Xatt, 'Xatt', X.dtype)
    seed, offset, flags = _rng_key(seed, offset, flags)
    rc = lib.apa_attn_pool_fwd_ex(
        _hooks_ptr(hooks), _dev_ptr(X, 'X'), xatt_ptr, _dev_ptr(Wa, 'Wa', torch.float32),
        _dev_ptr(ba, 'ba', torch.float32), _dev_ptr(Wt, 'Wt', torch.float32),
        _dev_ptr(bt, 'bt', torch.float32), logits.data_ptr(), att.data_ptr(),
        _dev_ptr(zsave, 'zsave'), _dev_ptr(abar, 'abar'), _dev_ptr(topdown, 'topdown'),
        workspace.data_ptr(), workspace.numel(), N, P, C, Ca, K, M, flags, float(keep_prob),
        int(seed), offset, dt, _stream_ptr())
    _check(rc, 'apa_attn_pool_fwd')
    return logits, att, zsave, abar, topdown, workspace


def attn_pool_bwd(X, Xatt, Wa, ba, Wt, bt, att, zsave, abar, G, *, flags=0, keep_prob=1.0, seed=0,
                  offset=0, workspace=None, out=None, dxatt_rank1=False, hooks=None):
    """dX, dXatt, dWa, dba, dWt, dbt = attn_pool_bwd(...).  `out` may supply preallocated
    (dX, dXatt, dWa, dba, dWt, dbt) buffers (e.g. views into a flat DP gradient bucket).
    `dxatt_rank1=True` (separate attention input, one bottom-up map): the second result is dZ, f32
    [N*P], with dXatt = dZ (x) Wa left to the consumer (APA_FLAG_DXATT_RANK1)."""
    lib = load_library()
    N = X.shape[0]
    C = X.shape[-1]
    P = X.numel() // (N * C)
    Ca = Xatt.shape[-1]
    M = Wa.shape[1]
    K = Wt.shape[1]
    dt = _feat_dtype(X)
    dev = X.device
    fused = Xatt is X
    if dxatt_rank1 and not fused:
        flags |= APA_FLAG_DXATT_RANK1
    if out is None:
        dX = torch.empty_like(X)
        if fused:
            dXatt = None
        elif dxatt_rank1:
            dXatt = torch.empty((N * P,), dtype=torch.float32, device=dev)
        else:
            dXatt = torch.empty_like(Xatt)
        dWa = torch.empty_like(Wa)
        dba = torch.empty_like(ba)
        dWt = torch.empty_like(Wt)
        dbt = torch.empty_like(bt)
    else:
        dX, dXatt, dWa, dba, dWt, dbt = out
    need = int(lib.apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
    xatt_ptr = _dev_ptr(X, 'X') if fused else _dev_ptr(Xatt, 'Xatt', X.dtype)
    seed, offset, flags = _rng_key(seed, offset, flags)
    rc = lib.apa_attn_pool_bwd_ex(
        _hooks_ptr(hooks), _dev_ptr(X, 'X'), xatt_ptr, _dev_ptr(Wa, 'Wa', torch.float32),
        _dev_ptr(ba, 'ba', torch.float32), _dev_ptr(Wt, 'Wt', torch.float32),
        _dev_ptr(bt, 'bt', torch.float32), _dev_ptr(att, 'att', torch.float32),
        _dev_ptr(zsave, 'zsave'), _dev_ptr(abar, 'abar'), _dev_ptr(G, 'G', torch.float32),
        _dev_ptr(dX, 'dX'), _dev_ptr(dXatt, 'dXatt'), _dev_ptr(dWa, 'dWa'), _dev_ptr(dba, 'dba'),
        _dev_ptr(dWt, 'dWt'), _dev_ptr(dbt, 'dbt'), workspace.data_ptr(), workspace.numel(), N, P,
        C, Ca, K, M, flags, float(keep_prob), int(seed), offset, dt, _stream_ptr())
    _check(rc, 'apa_attn_pool_bwd')
    return dX, dXatt, dWa, dba, dWt, dbt


class ApaConcatFeat(ctypes.Structure):
    """`apa_concat_feat` of include/apa.h (..._WITH_POSE_FEAT, nets_factory.py:289-295)."""
    _fields_ = [('Xext', c_void_p), ('J', c_int), ('zext', c_void_p), ('dXext', c_void_p)]


def attn_pool_fwd_cat(X, Xatt, Xext, Wa, ba, Wt, bt, *, flags=0, keep_prob=1.0, seed=0, offset=0,
                      workspace=None, hooks=None):
    """logits, att, zsave, abar, zext, workspace = attn_pool_fwd_cat(...): attentional pooling over
    concat(X, Xext) without forming it (include/apa.h: apa_attn_pool_fwd_cat).  Xext [N,P,J] (or
    [N,H,W,J]) f32; Wt is the full [C+J, K] td_weights; M == 1."""
    lib = load_library()
    N, C = X.shape[0], X.shape[-1]
    P = X.numel() // (N * C)
    Ca, M, K, J = Xatt.shape[-1], Wa.shape[1], Wt.shape[1], Xext.shape[-1]
    if Wt.shape[0] != C + J or Xext.numel() != N * P * J:
        raise ApaError('attn_pool_fwd_cat: Wt must be [C+J, K] and Xext [N,P,J]')
    dev = X.device
    logits = torch.empty((N, K), dtype=torch.float32, device=dev)
    att = torch.empty((N, P, M), dtype=torch.float32, device=dev)
    zsave = torch.empty((N, C), dtype=torch.float32, device=dev)
    abar = torch.empty((N,), dtype=torch.float32, device=dev)
    zext = torch.empty((N, J), dtype=torch.float32, device=dev)
    need = int(lib.apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
    cat = ApaConcatFeat(_dev_ptr(Xext, 'Xext', torch.float32), J, zext.data_ptr(), None)
    xatt_ptr = _dev_ptr(X, 'X') if Xatt is X else _dev_ptr(Xatt, 'Xatt', X.dtype)
    seed, offset, flags = _rng_key(seed, offset, flags)
    rc = lib.apa_attn_pool_fwd_cat(
        ctypes.addressof(cat), _hooks_ptr(hooks), _dev_ptr(X, 'X'), xatt_ptr, _dev_ptr(Wa, 'Wa', torch.float32),
        _dev_ptr(ba, 'ba', torch.float32), _dev_ptr(Wt, 'Wt', torch.float32),
        _dev_ptr(bt, 'bt', torch.float32), logits.data_ptr(), att.data_ptr(), zsave.data_ptr(),
        abar.data_ptr(), None, workspace.data_ptr(), workspace.numel(), N, P, C, Ca, K, M, flags,
        float(keep_prob), int(seed), offset, _feat_dtype(X), _stream_ptr())
    _check(rc, 'apa_attn_pool_fwd_cat')
    return logits, att, zsave, abar, zext, workspace


def attn_pool_bwd_cat(X, Xatt, Xext, zext, Wa, ba, Wt, bt, att, zsave, abar, G, *, flags=0, keep_prob=1.0,
                      seed=0, offset=0, workspace=None, hooks=None):
    """dX, dXatt, dXext, dWa, dba, dWt, dbt = attn_pool_bwd_cat(...); dWt is [C+J, K]."""
    lib = load_library()
    N, C = X.shape[0], X.shape[-1]
    P = X.numel() // (N * C)
    Ca, M, K, J = Xatt.shape[-1], Wa.shape[1], Wt.shape[1], Xext.shape[-1]
    dev = X.device
    fused = Xatt is X
    dX = torch.empty_like(X)
    dXatt = None if fused else torch.empty_like(Xatt)
    dXext = torch.empty_like(Xext)
    dWa, dba, dWt, dbt = torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wt), torch.empty_like(bt)
    need = int(lib.apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
    cat = ApaConcatFeat(_dev_ptr(Xext, 'Xext', torch.float32), J, _dev_ptr(zext, 'zext', torch.float32),
                        dXext.data_ptr())
    xatt_ptr = _dev_ptr(X, 'X') if fused else _dev_ptr(Xatt, 'Xatt', X.dtype)
    seed, offset, flags = _rng_key(seed, offset, flags)
    rc = lib.apa_attn_pool_bwd_cat(
        ctypes.addressof(cat), _hooks_ptr(hooks), _dev_ptr(X, 'X'), xatt_ptr, _dev_ptr(Wa, 'Wa', torch.float32),
        _dev_ptr(ba, 'ba', torch.float32), _dev_ptr(Wt, 'Wt', torch.float32),
        _dev_ptr(bt, 'bt', torch.float32), _dev_ptr(att, 'att', torch.float32), _dev_ptr(zsave, 'zsave'),
        _dev_ptr(abar, 'abar'), _dev_ptr(G, 'G', torch.float32), dX.data_ptr(), _dev_ptr(dXatt, 'dXatt'),
        dWa.data_ptr(), dba.data_ptr(), dWt.data_ptr(), dbt.data_ptr(), workspace.data_ptr(),
        workspace.numel(), N, P, C, Ca, K, M, flags, float(keep_prob), int(seed), offset, _feat_dtype(X),
        _stream_ptr())
    _check(rc, 'apa_attn_pool_bwd_cat')
    return dX, dXatt, dXext, dWa, dba, dWt, dbt


def dropout_mask(shape, keep_prob, seed, offset, device='cuda') -> torch.Tensor:
    """The exact {0,1} mask APA_FLAG_TRAIN applies to X (uint8, `shape` = X.shape)."""
    lib = load_library()
    mask = torch.empty(shape, dtype=torch.uint8, device=device)
    rc = lib.apa_dropout_mask(_dev_ptr(mask, 'mask'), mask.numel(), float(keep_prob), int(seed),
                              int(offset), _stream_ptr())
    _check(rc, 'apa_dropout_mask')
    return mask


# --------------------------------------------------------------------------------------------
# PoseLogits head (nets_factory.py:147-160)
# --------------------------------------------------------------------------------------------
def pose_head_fwd(X, W1, b1, W2, b2, workspace=None, out=None):
    """Ppre [..,Cp] (dtype of X), Pl [..,J] f32, workspace = pose_head_fwd(X [N,P,C] or [N,H,W,C], ...).
    `out=(Ppre, Pl)`: write into these caller-owned buffers (fixed addresses, e.g. for a HeadTrainStep
    bound to Ppre as its attention input)."""
    lib = load_library()
    N, C = X.shape[0], X.shape[-1]
    P = X.numel() // (N * C)
    Cp, J = W1.shape[1], W2.shape[1]
    dt = _feat_dtype(X)
    need = int(lib.apa_pose_head_workspace_bytes(N, P, C, Cp, J, dt))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=X.device)
    if out is not None:
        Ppre, Pl = out
        if Ppre.dtype != X.dtype or Ppre.numel() != N * P * Cp or Pl.dtype != torch.float32 or Pl.numel() != N * P * J:
            raise ApaError('pose_head_fwd: out=(Ppre, Pl) must be {} [N*P*{}] and float32 [N*P*{}]'.format(
                X.dtype, Cp, J))
        _dev_ptr(Ppre, 'Ppre', X.dtype)
        _dev_ptr(Pl, 'Pl', torch.float32)
    else:
        Ppre = torch.empty(tuple(X.shape[:-1]) + (Cp,), dtype=X.dtype, device=X.device)
        Pl = torch.empty(tuple(X.shape[:-1]) + (J,), dtype=torch.float32, device=X.device)
    rc = lib.apa_pose_head_fwd(
        _dev_ptr(X, 'X'), _dev_ptr(W1, 'W1', torch.float32), _dev_ptr(b1, 'b1', torch.float32),
        _dev_ptr(W2, 'W2', torch.float32), _dev_ptr(b2, 'b2', torch.float32), Ppre.data_ptr(),
        Pl.data_ptr(), workspace.data_ptr(), workspace.numel(), N, P, C, Cp, J, dt, _stream_ptr())
    _check(rc, 'apa_pose_head_fwd')
    return Ppre, Pl, workspace


def pose_head_bwd(X, W1, W2, Ppre, dPl, dPpre_ext, *, dX=None, accumulate_dX=False, workspace=None, ws_from_fwd=False,
                  ext_rank1=None):
    """dX, dW1, db1, dW2, db2 = pose_head_bwd(...).  dPl: pose-loss gradient [..,J] f32 or None;
    dPpre_ext: gradient from the attention branch [..,Cp] (dtype of X) or None.  With
    accumulate_dX the product dPpre.W1^T is ADDED to the given dX buffer.  `ext_rank1=(row, col)`:
    the attention-branch gradient in rank-1 form row [N*P] (x) col [Cp], both f32
    (apa_pose_head_bwd_rank1ext; dPpre_ext must then be None).  `ws_from_fwd=True`: `workspace` is
    the one pose_head_fwd returned and nothing has written to it since (APA_POSE_WS_FROM_FWD: the bf16
    copy of W1 in it is reused)."""
    lib = load_library()
    N, C = X.shape[0], X.shape[-1]
    P = X.numel() // (N * C)
    Cp, J = W1.shape[1], W2.shape[1]
    dt = _feat_dtype(X)
    need = int(lib.apa_pose_head_workspace_bytes(N, P, C, Cp, J, dt))
    if workspace is None or workspace.numel() < need:
        if ws_from_fwd:
            raise ApaError('ws_from_fwd needs the workspace of the forward call')
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=X.device)
    acc_flag = (1 if accumulate_dX else 0) | (2 if ws_from_fwd else 0)
    if dX is None:
        if accumulate_dX:
            raise ApaError('accumulate_dX needs an existing dX buffer')
        dX = torch.empty_like(X)
    dW1, db1 = torch.empty_like(W1), torch.empty((Cp,), dtype=torch.float32, device=X.device)
    dW2, db2 = torch.empty_like(W2), torch.empty((J,), dtype=torch.float32, device=X.device)
    if ext_rank1 is not None:
        if dPpre_ext is not None:
            raise ApaError('give the attention-branch gradient either as a tensor or in rank-1 form')
        row, col = ext_rank1
        if row.numel() != N * P or col.numel() != Cp:
            raise ApaError('ext_rank1: row must have N*P = {} and col Cp = {} elements'.format(N * P, Cp))
        rc = lib.apa_pose_head_bwd_rank1ext(
            _dev_ptr(X, 'X'), _dev_ptr(W1, 'W1', torch.float32), _dev_ptr(W2, 'W2', torch.float32),
            _dev_ptr(Ppre, 'Ppre', X.dtype), _dev_ptr(dPl, 'dPl', torch.float32),
            _dev_ptr(row, 'ext_row', torch.float32), _dev_ptr(col, 'ext_col', torch.float32),
            _dev_ptr(dX, 'dX', X.dtype), acc_flag, dW1.data_ptr(), db1.data_ptr(),
            dW2.data_ptr(), db2.data_ptr(), workspace.data_ptr(), workspace.numel(), N, P, C, Cp, J, dt,
            _stream_ptr())
        _check(rc, 'apa_pose_head_bwd_rank1ext')
        return dX, dW1, db1, dW2, db2
    rc = lib.apa_pose_head_bwd(
        _dev_ptr(X, 'X'), _dev_ptr(W1, 'W1', torch.float32), _dev_ptr(W2, 'W2', torch.float32),
        _dev_ptr(Ppre, 'Ppre', X.dtype), _dev_ptr(dPl, 'dPl', torch.float32),
        _dev_ptr(dPpre_ext, 'dPpre_ext', X.dtype), _dev_ptr(dX, 'dX', X.dtype),
        acc_flag, dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
        workspace.data_ptr(), workspace.numel(), N, P, C, Cp, J, dt, _stream_ptr())
    _check(rc, 'apa_pose_head_bwd')
    return dX, dW1, db1, dW2, db2


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
def softmax_xent_fwd_bwd(logits, labels, *, wt=1.0, grad_scale=1.0, want_grad=True,
                         want_probs=False, want_pred=False):
    """(loss_buf[1+N], G, probs, pred): src/loss.py:74-80 + eval.py:193-197 fused."""
    lib = load_library()
    N, K = logits.shape
    dev = logits.device
    loss = torch.empty((1 + N,), dtype=torch.float32, device=dev)
    G = torch.empty_like(logits) if want_grad else None
    probs = torch.empty_like(logits) if want_probs else None
    pred = torch.empty((N,), dtype=torch.int64, device=dev) if want_pred else None
    rc = lib.apa_softmax_xent_fwd_bwd(
        _dev_ptr(logits, 'logits', torch.float32), _dev_ptr(labels, 'labels', torch.int64),
        loss.data_ptr(), _dev_ptr(G, 'G'), _dev_ptr(probs, 'probs'), _dev_ptr(pred, 'pred'), N, K,
        float(wt), float(grad_scale), _stream_ptr())
    _check(rc, 'apa_softmax_xent_fwd_bwd')
    return loss, G, probs, pred


def pose_l2_loss_fwd_bwd(Pl, lbl, valid, *, wt=1.0, grad_scale=1.0, want_grad=True):
    """(loss[1], dPl): src/loss.py:29-70.  Pl/lbl [N,P,J] (or [N,H,W,J]) f32, valid [N,J] bool/uint8."""
    lib = load_library()
    N = Pl.shape[0]
    J = Pl.shape[-1]
    P = Pl.numel() // (N * J)
    dev = Pl.device
    # a bool tensor already is one 0/1 byte per element: reinterpret, do not launch a conversion
    v8 = (valid.view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8)).contiguous()
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dPl = torch.empty_like(Pl) if want_grad else None
    need = int(lib.apa_pose_l2_workspace_bytes(N, P, J))
    ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
    rc = lib.apa_pose_l2_loss_fwd_bwd(
        _dev_ptr(Pl, 'Pl', torch.float32), _dev_ptr(lbl, 'lbl', torch.float32),
        _dev_ptr(v8, 'valid'), loss.data_ptr(), _dev_ptr(dPl, 'dPl'), ws.data_ptr(), ws.numel(),
        N, P, J, float(wt), float(grad_scale), _stream_ptr())
    _check(rc, 'apa_pose_l2_loss_fwd_bwd')
    return loss, dPl


ACTION_LOSS_KINDS = {'l2': 1, 'multi-label': 2, 'multi-label-2': 3}


def action_loss_fwd_bwd(kind: str, logits, labels, *, wt=1.0, grad_scale=1.0, pos_weight=10.0, want_grad=True):
    """loss [1], G [N,K] for the action losses of src/loss.py:81-101 other than softmax cross-entropy
    ('l2': labels int64 [N]; 'multi-label' / 'multi-label-2': labels f32 [N,K] multi-hot)."""
    lib = load_library()
    N, K = logits.shape
    dev = logits.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    G = torch.empty((N, K), dtype=torch.float32, device=dev) if want_grad else None
    ldt = torch.int64 if kind == 'l2' else torch.float32
    rc = lib.apa_action_loss_fwd_bwd(ACTION_LOSS_KINDS[kind], _dev_ptr(logits, 'logits', torch.float32),
                                     _dev_ptr(labels, 'labels', ldt), loss.data_ptr(), _dev_ptr(G, 'G'), N, K,
                                     float(wt), float(grad_scale), float(pos_weight), _stream_ptr())
    _check(rc, 'apa_action_loss_fwd_bwd')
    return loss, G


def pose_sampled_loss_fwd_bwd(Pl, lbl, valid, uniform, *, wt=1.0, grad_scale=1.0, want_grad=True):
    """loss [1], dPl, mask = LOSS_FN_POSE_SAMPLED (src/loss.py:36-52); `uniform` = the caller's
    uniform [0,1) draws, same shape as Pl."""
    lib = load_library()
    N, J = Pl.shape[0], Pl.shape[-1]
    P = Pl.numel() // (N * J)
    dev = Pl.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dPl = torch.empty_like(Pl) if want_grad else None
    mask = torch.empty_like(Pl)
    v8 = valid.to(torch.uint8).contiguous()
    rc = lib.apa_pose_sampled_loss_fwd_bwd(_dev_ptr(Pl, 'Pl', torch.float32), _dev_ptr(lbl, 'lbl', torch.float32),
                                           _dev_ptr(v8, 'valid'), _dev_ptr(uniform, 'uniform', torch.float32),
                                           loss.data_ptr(), _dev_ptr(dPl, 'dPl'), mask.data_ptr(), N, P, J,
                                           float(wt), float(grad_scale), _stream_ptr())
    _check(rc, 'apa_pose_sampled_loss_fwd_bwd')
    return loss, dPl, mask


def resize_bilinear_tf1(img, out_h: int, out_w: int):
    """tf.image.resize_images (TF 1.1 legacy bilinear) of an f32 [N,h,w,C] device tensor."""
    lib = load_library()
    N, h, w, C = img.shape
    out = torch.empty((N, out_h, out_w, C), dtype=torch.float32, device=img.device)
    _check(lib.apa_resize_bilinear_tf1(_dev_ptr(img, 'img', torch.float32), out.data_ptr(), N, h, w, C,
                                       int(out_h), int(out_w), _stream_ptr()), 'apa_resize_bilinear_tf1')
    return out


# --------------------------------------------------------------------------------------------
# the reference's own custom ops (src/custom_ops/custom_ops_factory.py:20-47)
# --------------------------------------------------------------------------------------------
def pose_to_heatmap(pose_label, im_ht, im_wd, out_wd, out_channels=16, marker_wd_ratio=0.1,
                    do_gauss_blur=True) -> Tuple[np.ndarray, np.ndarray]:
    """Same call surface as custom_ops_factory.py:20-28 (`pose_to_heatmap(*args, **kwargs)`):
    returns (uint8 heatmap [out_ht,out_wd,out_channels] = float heatmap * 255 truncated,
    bool valid [out_channels]).  Host op, like the reference (DEVICE_CPU)."""
    lib = load_library()
    pose = np.ascontiguousarray(np.asarray(pose_label, dtype=np.int64).reshape(-1))
    out_ht = int(lib.apa_pose_to_heatmap_out_ht(int(im_ht), int(im_wd), int(out_wd)))
    if out_ht < 0:
        raise ApaError('apa_pose_to_heatmap_out_ht: bad image size')
    hm = np.zeros((out_ht, int(out_wd), int(out_channels)), dtype=np.float32)
    valid = np.zeros((int(out_channels),), dtype=np.uint8)
    rc = lib.apa_pose_to_heatmap(
        pose.ctypes.data_as(POINTER(c_int64)), pose.size, int(im_ht), int(im_wd), int(out_wd),
        int(out_channels), float(marker_wd_ratio), 1 if do_gauss_blur else 0,
        hm.ctypes.data_as(POINTER(c_float)), valid.ctypes.data_as(POINTER(c_uint8)))
    _check(rc, 'apa_pose_to_heatmap')
    return (hm * np.float32(255.0)).astype(np.uint8), valid.astype(bool)


def pose_to_heatmap_float(pose_label, im_ht, im_wd, out_wd, out_channels=16, marker_wd_ratio=0.1,
                          do_gauss_blur=True) -> Tuple[np.ndarray, np.ndarray]:
    """The raw op outputs (float heatmap in [0,1], bool valid) before the python wrapper's *255."""
    lib = load_library()
    pose = np.ascontiguousarray(np.asarray(pose_label, dtype=np.int64).reshape(-1))
    out_ht = int(lib.apa_pose_to_heatmap_out_ht(int(im_ht), int(im_wd), int(out_wd)))
    hm = np.zeros((max(out_ht, 0), int(out_wd), int(out_channels)), dtype=np.float32)
    valid = np.zeros((int(out_channels),), dtype=np.uint8)
    rc = lib.apa_pose_to_heatmap(
        pose.ctypes.data_as(POINTER(c_int64)), pose.size, int(im_ht), int(im_wd), int(out_wd),
        int(out_channels), float(marker_wd_ratio), 1 if do_gauss_blur else 0,
        hm.ctypes.data_as(POINTER(c_float)), valid.ctypes.data_as(POINTER(c_uint8)))
    _check(rc, 'apa_pose_to_heatmap')
    return hm, valid.astype(bool)


def spatial_mean_bwd(dz: torch.Tensor, shape, dtype) -> torch.Tensor:
    """dX [N,..,C] (`dtype`) = dz [N,C] / P broadcast over the spatial positions: backward of the global
    average pool of the attention-free head (cfg 001)."""
    lib = load_library()
    N, C = int(shape[0]), int(shape[-1])
    P = 1
    for d in shape[1:-1]:
        P *= int(d)
    dX = torch.empty(tuple(shape), dtype=dtype, device=dz.device)
    rc = lib.apa_spatial_mean_bwd(_dev_ptr(dz, 'dz', torch.float32), dX.data_ptr(), N, P, C, _feat_dtype(dX),
                                  _stream_ptr())
    _check(rc, 'apa_spatial_mean_bwd')
    return dX


def zero_out_channels(to_zero: torch.Tensor, channels: torch.Tensor) -> torch.Tensor:
    """custom_ops_factory.py:30-32 / zero_out_channels.cc: out[...,c] = channels[c] ? in : 0."""
    lib = load_library()
    C = to_zero.shape[-1]
    out = torch.empty_like(to_zero)
    ch = channels.to(torch.uint8).contiguous()
    rc = lib.apa_zero_out_channels(_dev_ptr(to_zero, 'to_zero', torch.float32), _dev_ptr(ch, 'channels'),
                                   out.data_ptr(), to_zero.numel() // C, C, _stream_ptr())
    _check(rc, 'apa_zero_out_channels')
    return out


def pose_label_replay_resize(hm_u8: np.ndarray, orig_hw, crop_info, whether_flip: bool, out_side: int,
                             eps: float = 1e-14) -> np.ndarray:
    """src/preprocess_pipeline.py:21-45 + :195-214 on one heat-map: replay crop
    (crop_info = [y, x, h, w] in the coordinates of the orig_hw = (H, W) image) and flip, /255,
    min-max normalise, legacy-bilinear resize to out_side^2.  Host op.  Returns f32 [S,S,J]."""
    lib = load_library()
    hm = np.ascontiguousarray(hm_u8, dtype=np.uint8)
    h, w, J = hm.shape
    out = np.zeros((out_side, out_side, J), dtype=np.float32)
    rc = lib.apa_pose_label_replay_resize(
        hm.ctypes.data_as(POINTER(c_uint8)), h, w, J, int(orig_hw[0]), int(orig_hw[1]),
        int(crop_info[0]), int(crop_info[1]), int(crop_info[2]), int(crop_info[3]),
        1 if whether_flip else 0, int(out_side), float(eps), out.ctypes.data_as(POINTER(c_float)))
    _check(rc, 'apa_pose_label_replay_resize')
    return out


# --------------------------------------------------------------------------------------------
# frame pooling (nets_factory.py:354-374)
# --------------------------------------------------------------------------------------------
def pose_labels_device(poses, geoms, out_wd=200, J=16, marker_wd_ratio=0.1, out_side=15, device='cuda'):
    """The whole label path of src/preprocess_pipeline.py:150-214 for a batch on the device.
    poses: list of int64 arrays (x, y, vis triples, n_people*J of them); geoms: list of
    (im_ht, im_wd, aug_ht, aug_wd, crop_y, crop_x, crop_h, crop_w, flip) -- im_* = the ORIGINAL image
    size (keypoint scaling, preprocess_pipeline.py:155-157), aug_* = preproc_info['image_shape'], the
    size after the aspect-preserving resize, on which crop_* was recorded (:29-36); a 7-tuple
    (im_ht, im_wd, crop...) means aug == im (no resize).  Returns (labels f32 [N,S,S,J] device,
    valid bool [N,J] device, status int32 [N] device)."""
    lib = load_library()
    N = len(poses)
    flat = [np.asarray(p, dtype=np.int64).reshape(-1) for p in poses]
    max_vals = max(max(f.size for f in flat), 3 * J)
    pose_h = np.full((N, max_vals), -1, dtype=np.int64)
    for i, f in enumerate(flat):
        pose_h[i, :f.size] = f
    dev = torch.device(device)
    pose_d = torch.from_numpy(pose_h).to(dev)
    nv_d = torch.tensor([f.size for f in flat], dtype=torch.int32, device=dev)
    g9 = []
    for g in geoms:
        g = [int(v) for v in g]
        if len(g) == 7:
            g = g[:2] + g[:2] + g[2:]
        if len(g) != 9:
            raise ValueError('geom needs 9 (or 7) integers, got {}'.format(len(g)))
        g9.append(g)
    geom_d = torch.tensor(g9, dtype=torch.int32, device=dev)
    labels = torch.empty((N, out_side, out_side, J), dtype=torch.float32, device=dev)
    valid = torch.empty((N, J), dtype=torch.uint8, device=dev)
    status = torch.empty((N,), dtype=torch.int32, device=dev)
    _check(lib.apa_pose_labels_device(pose_d.data_ptr(), nv_d.data_ptr(), geom_d.data_ptr(), N, max_vals,
                                      int(out_wd), int(J), float(marker_wd_ratio), int(out_side),
                                      labels.data_ptr(), valid.data_ptr(), status.data_ptr(), _stream_ptr()),
           'apa_pose_labels_device')
    return labels, valid.bool(), status


def frame_pool_fwd(logits, frames_per_video, w=None, b=None):
    """pooled [B,K], tatt [B*F] or None."""
    lib = load_library()
    BF, K = logits.shape
    B = BF // frames_per_video
    pooled = torch.empty((B, K), dtype=torch.float32, device=logits.device)
    tatt = torch.empty((BF,), dtype=torch.float32, device=logits.device) if w is not None else None
    rc = lib.apa_frame_pool_fwd(_dev_ptr(logits, 'logits', torch.float32), _dev_ptr(w, 'w', torch.float32),
                                _dev_ptr(b, 'b', torch.float32), pooled.data_ptr(), _dev_ptr(tatt, 'tatt'),
                                B, frames_per_video, K, _stream_ptr())
    _check(rc, 'apa_frame_pool_fwd')
    return pooled, tatt


def frame_pool_bwd(logits, frames_per_video, w, tatt, dpooled):
    """dlogits [B*F,K], dw [K] or None, db [1] or None."""
    lib = load_library()
    BF, K = logits.shape
    B = BF // frames_per_video
    dlogits = torch.empty_like(logits)
    dw = torch.empty((K,), dtype=torch.float32, device=logits.device) if w is not None else None
    db = torch.empty((1,), dtype=torch.float32, device=logits.device) if w is not None else None
    scratch = torch.empty((BF,), dtype=torch.float32, device=logits.device) if w is not None else None
    rc = lib.apa_frame_pool_bwd(_dev_ptr(logits, 'logits', torch.float32), _dev_ptr(w, 'w', torch.float32),
                                _dev_ptr(tatt, 'tatt'), _dev_ptr(dpooled, 'dpooled', torch.float32),
                                dlogits.data_ptr(), _dev_ptr(dw, 'dw'), _dev_ptr(db, 'db'),
                                _dev_ptr(scratch, 'scratch'), B, frames_per_video, K, _stream_ptr())
    _check(rc, 'apa_frame_pool_bwd')
    return dlogits, dw, db


class HeadTrainStep:
    """apa_attn_head_train_step bound to caller-owned buffers: forward, softmax cross-entropy and
    backward of the head as ONE foreign call per step (the reference runs the same sequence inside
    one `sess.run(train_op)`, src/train.py:529-566).  All marshalling happens once, here; `run()`
    costs one ctypes call plus the kernel launches, which keeps the host ahead of a ~55 us step.

    Outputs live in the attributes `logits [N,K]`, `att [N,P,M]`, `zsave`, `abar`, `loss [1+N]`,
    `G [N,K]`, and in the gradient buffers passed as `grads = (dX, dXatt, dWa, dba, dWt, dbt)`
    (e.g. views into a flat data-parallel bucket).  `offset`: int, or a 1-element int64 CUDA tensor
    (device-side dropout counter, advanced by the backward pass).  `dxatt_rank1=True` (separate attention
    input, M == 1): grads[1] is dZ, float32 [N*P] (APA_FLAG_DXATT_RANK1) -- the form
    pose_head_bwd(ext_rank1=(dZ, wa)) consumes."""

    def __init__(self, X, Xatt, Wa, ba, Wt, bt, labels, grads, *, flags=0, keep_prob=1.0, seed=0,
                 offset=0, loss_wt=1.0, grad_scale=1.0, workspace=None, hooks=None, dxatt_rank1=False,
                 weight_images=False, share_with=None):
        """`weight_images=True` (per-class maps, M == K): the padded / concatenated operand images of the weights
        are built ONCE in this step's workspace (`self.weight_image_maps` describes them) and every run() passes
        APA_FLAG_WEIGHT_IMAGES: the caller keeps them current -- `momentum_sgd_step(..., images=...)` /
        `deploy.MomentumSGD.attach_weight_images(step, ...)` in the optimiser's own launch, or
        `refresh_weight_images()` after any other change of the weights.
        `share_with` (another HeadTrainStep of the same shapes): this step is bound to ITS output tensors, workspace
        and weight images -- a training loop has one set of them; only X / dX differ between the bound steps
        (bench.py's rotating feature-map sets)."""
        self.lib = load_library()
        N, C = X.shape[0], X.shape[-1]
        P = X.numel() // (N * C)
        Ca, M, K = Xatt.shape[-1], Wa.shape[1], Wt.shape[1]
        dev = X.device
        fused = Xatt is X
        if dxatt_rank1:     # cfg 003: `dXatt` receives dZ, f32 [N*P]; dXatt = dZ (x) Wa is left to the consumer
            if fused or M != 1:
                raise ApaError('HeadTrainStep: dxatt_rank1 needs a separate attention input and M == 1')
            if grads[1].dtype != torch.float32 or grads[1].numel() != N * P:
                raise ApaError('HeadTrainStep: dxatt_rank1 wants grads[1] = dZ, float32 [N*P]')
            flags |= APA_FLAG_DXATT_RANK1
        dX, dXatt, dWa, dba, dWt, dbt = grads
        self.hooks = hooks            # default apa_hooks of run(); kept alive here
        if share_with is not None:
            o = share_with
            if tuple(o.logits.shape) != (N, K) or tuple(o.att.shape) != (N, P, M):
                raise ApaError('HeadTrainStep: share_with needs a step of the same shapes')
            self.logits, self.att, self.zsave, self.abar, self.loss, self.G = o.logits, o.att, o.zsave, o.abar, o.loss, o.G
            workspace = o.workspace
        else:
            self.logits = torch.empty((N, K), dtype=torch.float32, device=dev)
            self.att = torch.empty((N, P, M), dtype=torch.float32, device=dev)
            self.zsave = torch.empty((N, C) if M == 1 else (N, P, K), dtype=torch.float32, device=dev)
            self.abar = torch.empty((N,), dtype=torch.float32, device=dev) if M == 1 else None
            self.loss = torch.empty((1 + N,), dtype=torch.float32, device=dev)
            self.G = torch.empty((N, K), dtype=torch.float32, device=dev)
        need = int(self.lib.apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
        self.workspace = workspace
        self._keep = (X, Xatt, Wa, ba, Wt, bt, labels, grads, offset, seed)
        self.weight_image_maps = []
        if share_with is not None and share_with.weight_image_maps:
            self.weight_image_maps = share_with.weight_image_maps       # built there, in the shared workspace
            flags |= APA_FLAG_WEIGHT_IMAGES
        elif weight_images and M == K and M > 1:
            self.weight_image_maps = per_class_weight_images(Wa, ba, Wt, bt, workspace, N, P, X.dtype)
            if self.weight_image_maps:
                flags |= APA_FLAG_WEIGHT_IMAGES
        seed, off, flags = _rng_key(seed, offset, flags)     # the pointers below stay valid
        self._args = [
            _dev_ptr(X, 'X'), _dev_ptr(X, 'X') if fused else _dev_ptr(Xatt, 'Xatt', X.dtype),
            _dev_ptr(Wa, 'Wa', torch.float32), _dev_ptr(ba, 'ba', torch.float32),
            _dev_ptr(Wt, 'Wt', torch.float32), _dev_ptr(bt, 'bt', torch.float32),
            _dev_ptr(labels, 'labels', torch.int64), float(loss_wt), float(grad_scale),
            self.logits.data_ptr(), self.att.data_ptr(), self.zsave.data_ptr(), _dev_ptr(self.abar, 'abar'),
            self.loss.data_ptr(), self.G.data_ptr(), _dev_ptr(dX, 'dX', X.dtype),
            None if fused else _dev_ptr(dXatt, 'dXatt', torch.float32 if dxatt_rank1 else X.dtype),
            _dev_ptr(dWa, 'dWa', torch.float32),
            _dev_ptr(dba, 'dba', torch.float32), _dev_ptr(dWt, 'dWt', torch.float32),
            _dev_ptr(dbt, 'dbt', torch.float32), workspace.data_ptr(), workspace.numel(), N, P, C, Ca, K, M,
            flags, float(keep_prob), int(seed), off, _feat_dtype(X)]
        self._fn = self.lib.apa_attn_head_train_step_ex

    def refresh_weight_images(self) -> None:
        """Rebuild the operand images from the current weights (after load_state_dict, or an optimiser that does not
        rewrite them in its own launch)."""
        if self.weight_image_maps:
            X, _, Wa, ba, Wt, bt = self._keep[:6]
            N, C = X.shape[0], X.shape[-1]
            per_class_weight_images(Wa, ba, Wt, bt, self.workspace, N, X.numel() // (N * C), X.dtype)

    def rebind(self, X=None, labels=None, offset=None) -> None:
        """Point the bound step at another feature map / label tensor of the SAME shape and dtype (a training loop
        whose backbone hands over a fresh conv5 tensor every step) and / or at another dropout offset (int), without
        re-allocating outputs or workspace.  Fused attention input only (Xatt is X)."""
        keep = list(self._keep)
        if X is not None:
            if self._args[0] != self._args[1]:
                raise ApaError('HeadTrainStep.rebind: a separate attention input is bound')
            if X.shape != keep[0].shape or X.dtype != keep[0].dtype:
                raise ApaError('HeadTrainStep.rebind: same shape and dtype expected')
            self._args[0] = self._args[1] = _dev_ptr(X, 'X')
            keep[0] = keep[1] = X
        if labels is not None:
            self._args[6] = _dev_ptr(labels, 'labels', torch.int64)
            keep[6] = labels
        if offset is not None:
            if isinstance(keep[8], torch.Tensor):
                raise ApaError('HeadTrainStep.rebind: the step was bound to a device-side dropout counter')
            self._args[-2] = int(offset)
            keep[8] = int(offset)
        self._keep = tuple(keep)

    def run(self, stream: Optional[int] = None, hooks=None) -> None:
        """Enqueue one step on `stream` (a raw hipStream_t) or torch's current stream; `hooks`
        (an ApaHooks) overrides the instance's for this call."""
        h = self.hooks if hooks is None else hooks
        rc = self._fn(_hooks_ptr(h), *self._args, _stream_ptr() if stream is None else stream)
        if rc != 0:
            _check(rc, 'apa_attn_head_train_step')


class ApaPoseAttnStepIO(ctypes.Structure):
    """`apa_pose_attn_step_io` of include/apa.h (field order is the ABI)."""
    _fields_ = ([(n, c_void_p) for n in ('X', 'W1', 'b1', 'W2', 'b2', 'W1_bf16', 'W2T_bf16', 'Wa', 'ba', 'Wt', 'bt', 'labels',
                                          'pose_labels', 'pose_valid')] +
                [('action_wt', c_float), ('pose_wt', c_float), ('grad_scale', c_float)] +
                [(n, c_void_p) for n in ('Ppre', 'Pl', 'att', 'logits', 'zsave', 'abar', 'loss_action', 'loss_pose',
                                          'G', 'dPl', 'dZ', 'dX', 'dW1', 'db1', 'dW2', 'db2', 'dWa', 'dba', 'dWt',
                                          'dbt')] +
                [('ws_pool', c_void_p), ('ws_pool_bytes', c_size_t), ('ws_pose', c_void_p),
                 ('ws_pose_bytes', c_size_t)])


class PoseAttnTrainStep:
    """apa_pose_attn_train_step bound to caller-owned buffers: the whole cfg 003 head step (PoseLogits head ->
    attention from pose_pre_logits -> dropout + pooling -> pose L2 + softmax cross-entropy -> every gradient) as ONE
    foreign call, like the reference's one sess.run per step (src/train.py:529-566).

    `params = (W1, b1, W2, b2, Wa, ba, Wt, bt)` fp32; `grads = (dX, dW1, db1, dW2, db2, dWa, dba, dWt, dbt)`;
    `w1_bf16`: optional bf16 copy of W1 the caller keeps current (momentum_sgd_step(..., shadows=...)).
    Outputs as attributes: Ppre, Pl, att, logits, zsave, abar, loss_action [1+N], loss_pose [1], G, dPl, dZ."""

    def __init__(self, X, params, labels, pose_labels, pose_valid, grads, *, flags=0, keep_prob=1.0, seed=0,
                 offset=0, action_wt=1.0, pose_wt=1.0, grad_scale=1.0, w1_bf16=None, share_with=None,
                 w2t_bf16=None):
        """`share_with`: another PoseAttnTrainStep of the same shapes whose activation / loss buffers and workspaces
        this step is bound to as well (see HeadTrainStep).  `w2t_bf16`: optional bf16 [16, Cp + 16] image of W2^T (rows
        J.. and the pad columns zero) the caller keeps current -- `pose_w2t_image(W2)` builds it, `pose_w2t_image_map(image, W2)` describes
        it for `momentum_sgd_step(..., images=...)`."""
        self.lib = load_library()
        W1, b1, W2, b2, Wa, ba, Wt, bt = params
        dX, dW1, db1, dW2, db2, dWa, dba, dWt, dbt = grads
        N, C = X.shape[0], X.shape[-1]
        P = X.numel() // (N * C)
        Cp, J, K = W1.shape[1], W2.shape[1], Wt.shape[1]
        dev, f32 = X.device, torch.float32
        if tuple(W1.shape) != (C, Cp) or Wa.numel() != Cp or tuple(Wt.shape) != (C, K):
            raise ApaError('PoseAttnTrainStep: W1 [C,Cp], Wa [Cp,1], Wt [C,K] expected')
        if pose_valid.dtype == torch.bool:
            pose_valid = pose_valid.to(torch.uint8)
        new = lambda *shape, dt=f32: torch.empty(shape, dtype=dt, device=dev)
        _shared = ('Ppre', 'Pl', 'att', 'logits', 'zsave', 'abar', 'loss_action', 'loss_pose', 'G', 'dPl', 'dZ')
        if share_with is not None:
            if tuple(share_with.Ppre.shape) != (N, P, Cp) or share_with.Ppre.dtype != X.dtype or \
                    tuple(share_with.logits.shape) != (N, K):
                raise ApaError('PoseAttnTrainStep: share_with needs a step of the same shapes')
            for name in _shared:
                setattr(self, name, getattr(share_with, name))
        else:
            self.Ppre, self.Pl, self.att = new(N, P, Cp, dt=X.dtype), new(N, P, J), new(N, P, 1)
            self.logits, self.zsave, self.abar = new(N, K), new(N, C), new(N)
            self.loss_action, self.loss_pose = new(1 + N), new(1)
            self.G, self.dPl, self.dZ = new(N, K), new(N, P, J), new(N * P)
        dt = _feat_dtype(X)
        seed, off, flags = _rng_key(seed, offset, flags)
        if flags & APA_FLAG_RNG_EXTERNAL:
            raise ApaError('PoseAttnTrainStep: an external keep mask is served by the per-op entry points')
        if share_with is not None:
            self.ws_pool, self.ws_pose = share_with.ws_pool, share_with.ws_pose
        else:
            self.ws_pool = torch.empty((max(int(self.lib.apa_attn_pool_workspace_bytes(N, P, C, Cp, K, 1, flags)), 16),),
                                       dtype=torch.uint8, device=dev)
            self.ws_pose = torch.empty((max(int(self.lib.apa_pose_head_workspace_bytes(N, P, C, Cp, J, dt)), 16),),
                                       dtype=torch.uint8, device=dev)
        self._keep = (X, params, labels, pose_labels, pose_valid, grads, offset, w1_bf16, w2t_bf16)
        io = ApaPoseAttnStepIO()
        io.X = _dev_ptr(X, 'X')
        for name, t in (('W1', W1), ('b1', b1), ('W2', W2), ('b2', b2), ('Wa', Wa), ('ba', ba), ('Wt', Wt), ('bt', bt),
                        ('pose_labels', pose_labels), ('dW1', dW1), ('db1', db1), ('dW2', dW2), ('db2', db2),
                        ('dWa', dWa), ('dba', dba), ('dWt', dWt), ('dbt', dbt)):
            setattr(io, name, _dev_ptr(t, name, f32))
        io.W1_bf16 = None if w1_bf16 is None else _dev_ptr(w1_bf16, 'w1_bf16', torch.bfloat16)
        io.W2T_bf16 = None
        if w2t_bf16 is not None:
            if tuple(w2t_bf16.shape) != (16, Cp + 16) or J > 16 or not w2t_bf16.is_contiguous():
                raise ApaError('PoseAttnTrainStep: w2t_bf16 is a contiguous bf16 [16, Cp + 16] image of W2^T (J <= 16; '
                               'pose_w2t_image)')
            io.W2T_bf16 = _dev_ptr(w2t_bf16, 'w2t_bf16', torch.bfloat16)
        if w1_bf16 is not None and w1_bf16.numel() != W1.numel():
            raise ApaError('PoseAttnTrainStep: w1_bf16 must have W1\'s element count')
        io.labels = _dev_ptr(labels, 'labels', torch.int64)
        io.pose_valid = _dev_ptr(pose_valid, 'pose_valid', torch.uint8)
        if pose_labels.numel() != N * P * J or pose_valid.numel() != N * J or labels.numel() != N:
            raise ApaError('PoseAttnTrainStep: labels [N], pose_labels [N,P,J], pose_valid [N,J] expected')
        io.action_wt, io.pose_wt, io.grad_scale = float(action_wt), float(pose_wt), float(grad_scale)
        io.dX = _dev_ptr(dX, 'dX', X.dtype)
        for name in ('Ppre', 'Pl', 'att', 'logits', 'zsave', 'abar', 'loss_action', 'loss_pose', 'G', 'dPl', 'dZ'):
            setattr(io, name, getattr(self, name).data_ptr())
        io.ws_pool, io.ws_pool_bytes = self.ws_pool.data_ptr(), self.ws_pool.numel()
        io.ws_pose, io.ws_pose_bytes = self.ws_pose.data_ptr(), self.ws_pose.numel()
        self._io = io
        self._args = [ctypes.addressof(io), N, P, C, Cp, J, K, flags, float(keep_prob), int(seed), off, dt]

    def rebind(self, X=None, labels=None, pose_labels=None, pose_valid=None, offset=None) -> None:
        """Point the bound step at other inputs of the same shapes / dtypes and / or another dropout offset (int)
        without re-allocating outputs or workspaces (see HeadTrainStep.rebind)."""
        keep = list(self._keep)
        io = self._io
        if X is not None:
            if X.shape != keep[0].shape or X.dtype != keep[0].dtype:
                raise ApaError('PoseAttnTrainStep.rebind: same shape and dtype expected')
            io.X = _dev_ptr(X, 'X')
            keep[0] = X
        if labels is not None:
            io.labels = _dev_ptr(labels, 'labels', torch.int64)
            keep[2] = labels
        if pose_labels is not None:
            if pose_labels.numel() != keep[3].numel():
                raise ApaError('PoseAttnTrainStep.rebind: pose_labels [N,P,J] expected')
            io.pose_labels = _dev_ptr(pose_labels, 'pose_labels', torch.float32)
            keep[3] = pose_labels
        if pose_valid is not None:
            if pose_valid.dtype == torch.bool:
                pose_valid = pose_valid.to(torch.uint8)
            io.pose_valid = _dev_ptr(pose_valid, 'pose_valid', torch.uint8)
            keep[4] = pose_valid
        if offset is not None:
            if isinstance(keep[6], torch.Tensor):
                raise ApaError('PoseAttnTrainStep.rebind: the step was bound to a device-side dropout counter')
            self._args[10] = int(offset)
            keep[6] = int(offset)
        self._keep = tuple(keep)

    def run(self, stream: Optional[int] = None) -> None:
        rc = self.lib.apa_pose_attn_train_step(*self._args, _stream_ptr() if stream is None else stream)
        if rc != 0:
            _check(rc, 'apa_pose_attn_train_step')


class HeadEvalStep:
    """apa_attn_head_eval_step bound to caller-owned inputs: forward + softmax probabilities + argmax
    (+ per-example loss when `labels` is given) as ONE foreign call (eval.py:181-197).  Outputs:
    `logits`, `att`, `probs` [N,K], `pred` [N] int64, `loss` [1+N] or None."""

    def __init__(self, X, Xatt, Wa, ba, Wt, bt, labels=None, *, flags=0, workspace=None):
        self.lib = load_library()
        N, C = X.shape[0], X.shape[-1]
        P = X.numel() // (N * C)
        Ca, M, K = Xatt.shape[-1], Wa.shape[1], Wt.shape[1]
        dev = X.device
        flags &= ~APA_FLAG_TRAIN
        self.logits = torch.empty((N, K), dtype=torch.float32, device=dev)
        self.att = torch.empty((N, P, M), dtype=torch.float32, device=dev)
        self.zsave = torch.empty((N, C) if M == 1 else (N, P, K), dtype=torch.float32, device=dev)
        self.abar = torch.empty((N,), dtype=torch.float32, device=dev) if M == 1 else None
        self.probs = torch.empty((N, K), dtype=torch.float32, device=dev)
        self.pred = torch.empty((N,), dtype=torch.int64, device=dev)
        self.loss = torch.empty((1 + N,), dtype=torch.float32, device=dev) if labels is not None else None
        need = int(self.lib.apa_attn_pool_workspace_bytes(N, P, C, Ca, K, M, flags))
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=dev)
        self.workspace = workspace
        self._keep = (X, Xatt, Wa, ba, Wt, bt, labels)
        self._args = [
            _dev_ptr(X, 'X'), _dev_ptr(X, 'X') if Xatt is X else _dev_ptr(Xatt, 'Xatt', X.dtype),
            _dev_ptr(Wa, 'Wa', torch.float32), _dev_ptr(ba, 'ba', torch.float32),
            _dev_ptr(Wt, 'Wt', torch.float32), _dev_ptr(bt, 'bt', torch.float32),
            _dev_ptr(labels, 'labels', torch.int64), self.logits.data_ptr(), self.att.data_ptr(),
            self.zsave.data_ptr(), _dev_ptr(self.abar, 'abar'), _dev_ptr(self.loss, 'loss'),
            self.probs.data_ptr(), self.pred.data_ptr(), workspace.data_ptr(), workspace.numel(),
            N, P, C, Ca, K, M, flags, _feat_dtype(X)]
        self._fn = self.lib.apa_attn_head_eval_step

    def run(self, stream: Optional[int] = None) -> None:
        rc = self._fn(*self._args, _stream_ptr() if stream is None else stream)
        if rc != 0:
            _check(rc, 'apa_attn_head_eval_step')


class ApaWeightImage(ctypes.Structure):
    """`apa_weight_image` of include/apa.h (field order is the ABI): where the optimiser's launch stores the updated
    element (c, k) of one head parameter inside a per-class workspace."""
    _fields_ = [('dst', c_void_p), ('role', c_int), ('is_f32', c_int), ('cols', c_int), ('c_shift', c_int),
                ('a', c_int), ('b', c_int), ('d', c_int), ('e', c_int)]


def per_class_weight_images(Wa, ba, Wt, bt, workspace, N, P, dtype):
    """apa_per_class_weight_images: build, inside `workspace`, every padded / concatenated operand image the
    per-class head (M == K) would otherwise rebuild in each call, and return their descriptions
    [(role 'Wa'|'ba'|'Wt'|'bt', ApaWeightImage), ...] for `momentum_sgd_step(..., images=...)`.  Calls that then
    pass APA_FLAG_WEIGHT_IMAGES with this workspace skip the per-step weight preparation.  `dtype`: the FEATURE
    dtype the images are for (torch.bfloat16 / torch.float32)."""
    lib = load_library()
    Ca, K = Wa.shape
    C = Wt.shape[0]
    if tuple(Wt.shape) != (C, K) or ba.numel() != K or bt.numel() != K:
        raise ApaError('per_class_weight_images: Wa [Ca,K], ba [K], Wt [C,K], bt [K] expected (M == K)')
    maps = (ApaWeightImage * APA_WIMG_MAX)()
    n = c_int(0)
    dt = APA_DTYPE_BF16 if dtype == torch.bfloat16 else APA_DTYPE_F32
    _check(lib.apa_per_class_weight_images(
        _dev_ptr(Wa, 'Wa', torch.float32), _dev_ptr(ba, 'ba', torch.float32), _dev_ptr(Wt, 'Wt', torch.float32),
        _dev_ptr(bt, 'bt', torch.float32), workspace.data_ptr(), workspace.numel(), N, P, C, Ca, K, dt,
        ctypes.addressof(maps), ctypes.byref(n), _stream_ptr()), 'apa_per_class_weight_images')
    out = []
    for i in range(n.value):
        m = ApaWeightImage()
        ctypes.memmove(ctypes.addressof(m), ctypes.addressof(maps[i]), ctypes.sizeof(ApaWeightImage))
        out.append((APA_WIMG_ROLES[m.role], m))
    return out


def pose_w2t_image(W2: torch.Tensor) -> torch.Tensor:
    """bf16 [16, Cp + 16] image of W2^T (W2 [Cp, J <= 16]; rows J.. and the 16 pad columns zero) -- the LDS image of
    the Pl product, which copies it verbatim: `PoseAttnTrainStep(w2t_bf16=...)`."""
    Cp, J = W2.shape
    img = torch.zeros((16, Cp + 16), dtype=torch.bfloat16, device=W2.device)
    img[:J, :Cp].copy_(W2.t())
    return img


def pose_w2t_image_map(image: torch.Tensor, W2: torch.Tensor) -> ApaWeightImage:
    """where the optimiser's launch stores the updated element (c, j) of W2 inside `image`: image[j, c]."""
    m = ApaWeightImage()
    m.dst, m.role, m.is_f32, m.cols, m.c_shift = image.data_ptr(), 0, 0, int(W2.shape[1]), 0
    m.a, m.b, m.d, m.e = 1, 0, int(image.shape[1]), 0
    return m


def momentum_sgd_step(weights, weight_decay, grad_flat, acc_flat, lr, momentum=0.9, grad_scale=1.0, shadows=None,
                      images=None):
    """One fused launch: acc = m*acc + (grad_scale*g + wd_i*w_i); w_i -= lr*acc for every parameter.
    `weights`: list of fp32 device tensors in bucket order; `weight_decay`: one float per tensor.
    `shadows`: optional list (one entry per tensor, None = no shadow) of bf16 device tensors that receive the
    UPDATED weights rounded to bf16 in the same launch (apa_momentum_sgd_step_shadow: the pose head's W1)."""
    lib = load_library()
    n = len(weights)
    ptrs = (c_void_p * n)(*[_dev_ptr(w, 'weights[%d]' % i, torch.float32) for i, w in enumerate(weights)])
    sizes = (c_size_t * n)(*[w.numel() for w in weights])
    wds = (c_float * n)(*[float(x) for x in weight_decay])
    total = sum(w.numel() for w in weights)
    if grad_flat.numel() != total or acc_flat.numel() != total:
        raise ValueError('flat buffers hold {} / {} elements, parameters {}'.format(
            grad_flat.numel(), acc_flat.numel(), total))
    if images:
        # `images`: [(segment index, ApaWeightImage), ...] -- the per-class head's operand images, rewritten from the
        # updated weights by this same launch (apa_momentum_sgd_step_images)
        sh = None
        if shadows is not None and any(t is not None for t in shadows):
            sh = (c_void_p * n)(*[None if t is None else _dev_ptr(t, 'shadow', torch.bfloat16) for t in shadows])
        ni = len(images)
        arr = (ApaWeightImage * ni)()
        seg = (c_int * ni)(*[int(i) for i, _ in images])
        for q, (_, m) in enumerate(images):
            ctypes.memmove(ctypes.addressof(arr[q]), ctypes.addressof(m), ctypes.sizeof(ApaWeightImage))
        _check(lib.apa_momentum_sgd_step_images(
            n, ptrs, sizes, wds, _dev_ptr(grad_flat, 'grad_flat', torch.float32),
            _dev_ptr(acc_flat, 'acc_flat', torch.float32), lr, momentum, grad_scale, sh, ctypes.addressof(arr), seg, ni,
            _stream_ptr()), 'apa_momentum_sgd_step_images')
        return
    if shadows is not None and any(t is not None for t in shadows):
        for w_, t in zip(weights, shadows):
            if t is not None and (t.dtype != torch.bfloat16 or t.numel() != w_.numel() or not t.is_contiguous()):
                raise ApaError('momentum_sgd_step: a shadow must be a contiguous bf16 tensor of its weight\'s size')
        sh = (c_void_p * n)(*[None if t is None else _dev_ptr(t, 'shadow', torch.bfloat16) for t in shadows])
        _check(lib.apa_momentum_sgd_step_shadow(
            n, ptrs, sizes, wds, _dev_ptr(grad_flat, 'grad_flat', torch.float32),
            _dev_ptr(acc_flat, 'acc_flat', torch.float32), lr, momentum, grad_scale, sh, _stream_ptr()),
            'apa_momentum_sgd_step_shadow')
        return
    _check(lib.apa_momentum_sgd_step(n, ptrs, sizes, wds, _dev_ptr(grad_flat, 'grad_flat', torch.float32),
                                     _dev_ptr(acc_flat, 'acc_flat', torch.float32), lr, momentum,
                                     grad_scale, _stream_ptr()), 'apa_momentum_sgd_step')


class BoundMomentumSGD:
    """`momentum_sgd_step` with the marshalling done ONCE (the optimiser's counterpart of HeadTrainStep): the pointer /
    size / decay arrays, the shadow table and the image maps are built here, `run(lr, momentum, grad_scale)` is one
    foreign call.  The per-update python cost of the unbound form (~40 us of ctypes array building for eight
    parameters) is of the order of a whole HMDB-51 head step; a training loop pays it once per binding."""

    def __init__(self, weights, weight_decay, grad_flat, acc_flat, shadows=None, images=None):
        self.lib = load_library()
        n = len(weights)
        total = sum(w.numel() for w in weights)
        if grad_flat.numel() != total or acc_flat.numel() != total:
            raise ValueError('flat buffers hold {} / {} elements, parameters {}'.format(
                grad_flat.numel(), acc_flat.numel(), total))
        self._keep = (list(weights), grad_flat, acc_flat, list(shadows) if shadows else None,
                      list(images) if images else None)
        self._n = n
        self._ptrs = (c_void_p * n)(*[_dev_ptr(w, 'weights[%d]' % i, torch.float32) for i, w in enumerate(weights)])
        self._sizes = (c_size_t * n)(*[w.numel() for w in weights])
        self._wds = (c_float * n)(*[float(x) for x in weight_decay])
        self._g = _dev_ptr(grad_flat, 'grad_flat', torch.float32)
        self._a = _dev_ptr(acc_flat, 'acc_flat', torch.float32)
        self._sh = None
        if shadows is not None and any(t is not None for t in shadows):
            for w_, t in zip(weights, shadows):
                if t is not None and (t.dtype != torch.bfloat16 or t.numel() != w_.numel() or not t.is_contiguous()):
                    raise ApaError('momentum_sgd_step: a shadow must be a contiguous bf16 tensor of its weight\'s size')
            self._sh = (c_void_p * n)(*[None if t is None else _dev_ptr(t, 'shadow', torch.bfloat16) for t in shadows])
        self._ni = 0
        if images:
            ni = len(images)
            self._arr = (ApaWeightImage * ni)()
            self._seg = (c_int * ni)(*[int(i) for i, _ in images])
            for q, (_, m) in enumerate(images):
                ctypes.memmove(ctypes.addressof(self._arr[q]), ctypes.addressof(m), ctypes.sizeof(ApaWeightImage))
            self._ni = ni

    def run(self, lr, momentum=0.9, grad_scale=1.0, stream: Optional[int] = None) -> None:
        st = _stream_ptr() if stream is None else stream
        if self._ni:
            rc = self.lib.apa_momentum_sgd_step_images(self._n, self._ptrs, self._sizes, self._wds, self._g, self._a, lr,
                                                       momentum, grad_scale, self._sh, ctypes.addressof(self._arr),
                                                       self._seg, self._ni, st)
            who = 'apa_momentum_sgd_step_images'
        elif self._sh is not None:
            rc = self.lib.apa_momentum_sgd_step_shadow(self._n, self._ptrs, self._sizes, self._wds, self._g, self._a, lr,
                                                       momentum, grad_scale, self._sh, st)
            who = 'apa_momentum_sgd_step_shadow'
        else:
            rc = self.lib.apa_momentum_sgd_step(self._n, self._ptrs, self._sizes, self._wds, self._g, self._a, lr,
                                                momentum, grad_scale, st)
            who = 'apa_momentum_sgd_step'
        if rc != 0:
            _check(rc, who)


def _optim_segments(weights, weight_decay, flats, shadows, who):
    n = len(weights)
    ptrs = (c_void_p * n)(*[_dev_ptr(w, 'weights[%d]' % i, torch.float32) for i, w in enumerate(weights)])
    sizes = (c_size_t * n)(*[w.numel() for w in weights])
    wds = (c_float * n)(*[float(x) for x in weight_decay])
    total = sum(w.numel() for w in weights)
    for f in flats:
        if f.numel() != total:
            raise ValueError('%s: a flat buffer holds %d elements, the parameters %d' % (who, f.numel(), total))
    sh = None
    if shadows is not None and any(t is not None for t in shadows):
        for w_, t in zip(weights, shadows):
            if t is not None and (t.dtype != torch.bfloat16 or t.numel() != w_.numel() or not t.is_contiguous()):
                raise ApaError('%s: a shadow must be a contiguous bf16 tensor of its weight\'s size' % who)
        sh = (c_void_p * n)(*[None if t is None else _dev_ptr(t, 'shadow', torch.bfloat16) for t in shadows])
    return n, ptrs, sizes, wds, sh


def adam_step(weights, weight_decay, grad_flat, m_flat, v_flat, lr, t, beta1=0.9, beta2=0.999, epsilon=1e-8,
              grad_scale=1.0, shadows=None):
    """tf.train.AdamOptimizer (src/train.py:84-89) as one fused launch; `t` = number of this update (1, 2, ...):
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) is formed here, in double precision, like TF's python side."""
    lib = load_library()
    n, ptrs, sizes, wds, sh = _optim_segments(weights, weight_decay, (grad_flat, m_flat, v_flat), shadows, 'adam_step')
    lr_t = float(lr) * (1.0 - float(beta2) ** int(t)) ** 0.5 / (1.0 - float(beta1) ** int(t))
    _check(lib.apa_adam_step(n, ptrs, sizes, wds, _dev_ptr(grad_flat, 'grad_flat', torch.float32),
                             _dev_ptr(m_flat, 'm_flat', torch.float32), _dev_ptr(v_flat, 'v_flat', torch.float32),
                             lr_t, beta1, beta2, epsilon, grad_scale, sh, _stream_ptr()), 'apa_adam_step')


def rmsprop_step(weights, weight_decay, grad_flat, ms_flat, mom_flat, lr, decay=0.9, momentum=0.0, epsilon=1e-10,
                 grad_scale=1.0, shadows=None):
    """tf.train.RMSPropOptimizer (src/train.py:95-100) as one fused launch; `ms_flat` starts at ONE."""
    lib = load_library()
    n, ptrs, sizes, wds, sh = _optim_segments(weights, weight_decay, (grad_flat, ms_flat, mom_flat), shadows,
                                              'rmsprop_step')
    _check(lib.apa_rmsprop_step(n, ptrs, sizes, wds, _dev_ptr(grad_flat, 'grad_flat', torch.float32),
                                _dev_ptr(ms_flat, 'ms_flat', torch.float32),
                                _dev_ptr(mom_flat, 'mom_flat', torch.float32), lr, decay, momentum, epsilon,
                                grad_scale, sh, _stream_ptr()), 'apa_rmsprop_step')


# --------------------------------------------------------------------------------------------
# measurement hooks (bench.py)
# --------------------------------------------------------------------------------------------
ACC_MAX_PARTS = 8      # APA_ACC_MAX_PARTS (include/apa.h)


def accumulate_gradients(out: torch.Tensor, parts, scale: Optional[float] = None, stream: Optional[int] = None,
                         divisor: Optional[float] = None) -> None:
    """out = (parts[0] + parts[1] + ...) / divisor  -- or  ... * scale --, summed in that order (TRAIN.ITER_SIZE
    accumulation of micro-batch buckets, src/train.py:529-566).  `divisor=ITER_SIZE` is the reference's own
    arithmetic (`ref_grad / float(ITER_SIZE)`, apa_accumulate_gradients_div).  More than APA_ACC_MAX_PARTS
    parts are folded in groups, left to right, so the summation order stays that of the sequential loop."""
    if (scale is None) == (divisor is None):
        raise ApaError('accumulate_gradients: give exactly one of scale / divisor')
    lib = load_library()
    n = out.numel()
    parts = list(parts)
    for p_ in parts:
        if p_.numel() != n:
            raise ApaError('accumulate_gradients: every part must have out.numel() elements')
    st = _stream_ptr() if stream is None else stream
    optr = _dev_ptr(out, 'out', torch.float32)

    def launch(group, last):
        arr = (c_void_p * len(group))(*[_dev_ptr(p_, 'part', torch.float32) for p_ in group])
        if last and divisor is not None:
            rc = lib.apa_accumulate_gradients_div(optr, arr, len(group), n, float(divisor), st)
        else:
            rc = lib.apa_accumulate_gradients(optr, arr, len(group), n, float(scale) if last else 1.0, st)
        _check(rc, 'apa_accumulate_gradients')

    head, rest = parts[:ACC_MAX_PARTS], parts[ACC_MAX_PARTS:]
    launch(head, not rest)
    while rest:                               # out already holds the running sum: it leads the next group
        group, rest = [out] + rest[:ACC_MAX_PARTS - 1], rest[ACC_MAX_PARTS - 1:]
        launch(group, not rest)


class KernelTimer:
    """Per-step event pairs for the two streaming kernels of the M == 1 head (m1s_pool_fwd_kernel,
    m1s_bwd_main_kernel).  `hooks(i)` is the apa_hooks struct to pass to step i: the library launches
    those kernels through hipExtLaunchKernel with the pair attached, so each pair brackets exactly one
    dispatch on its own stream (reads ~1.2 us above rocprofv3's begin -> end of the same kernel)."""

    def __init__(self, n_steps: int, base: Optional[ApaHooks] = None):
        lib = load_library()
        self._lib = lib
        self.fwd, self.bwd, self._hooks = [], [], []
        for _ in range(n_steps):
            f, b = self._new_pair(), self._new_pair()
            self.fwd.append(f)
            self.bwd.append(b)
            h = make_hooks(prof_fwd=f, prof_bwd=b)
            if base is not None:
                h.grad_ready_event, h.td_weights_ready_event = base.grad_ready_event, base.td_weights_ready_event
            self._hooks.append(h)

    def _new_pair(self):
        a, b = c_void_p(), c_void_p()
        _check(self._lib.apa_prof_event_create(ctypes.byref(a)), 'apa_prof_event_create')
        _check(self._lib.apa_prof_event_create(ctypes.byref(b)), 'apa_prof_event_create')
        return a, b

    def hooks(self, i: int) -> ApaHooks:
        return self._hooks[i]

    def _elapsed(self, pairs):
        out = []
        for a, b in pairs:
            ms = c_float()
            _check(self._lib.apa_prof_event_elapsed_ms(a, b, ctypes.byref(ms)), 'apa_prof_event_elapsed_ms')
            out.append(ms.value)
        return out

    def fwd_elapsed_ms(self):
        return self._elapsed(self.fwd)

    def bwd_elapsed_ms(self):
        return self._elapsed(self.bwd)

    def close(self) -> None:
        for a, b in self.fwd + self.bwd:
            self._lib.apa_prof_event_destroy(a)
            self._lib.apa_prof_event_destroy(b)
        self.fwd, self.bwd, self._hooks = [], [], []
