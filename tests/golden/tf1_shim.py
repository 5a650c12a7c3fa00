"""A minimal float64 stand-in for the parts of TensorFlow 1.1 / tf.contrib.slim that the reference's head
code calls  --  TEST INFRASTRUCTURE, used only by tests/golden/make_head_reference.py.

Purpose: /root/reference/models/slim/nets/nets_factory.py (get_network_fn / network_fn, :94-380) and
/root/reference/src/loss.py (gen_losses, :4-105) are plain Python graph-construction code.  With this
module installed as `tensorflow`, those two files are EXEC'D FROM THE REFERENCE TREE and the reference's
own statements -- its scopes, its flag tests, its chained `net = slim.conv2d(net, ...)`, its lbl/lgt swap,
its reduce_mean / reduce_sum choices, its arg_scope defaults leaking into un-annotated convs -- decide what
is computed.  What this file supplies is only the meaning of each individual TF op, written from the TF 1.1
sources' documented behaviour and NOT from oracle/attn_pool_oracle.py, which the
fixtures are meant to check:

    op (call site)                                   semantics implemented here
    ------------------------------------------------ -------------------------------------------------------
    slim.arg_scope / add_arg_scope                    scope stack of per-op keyword defaults; a dict argument
                                                      REPLACES the current scope (arg_scope.py)
    tf.variable_scope(name, default_name)             name stack; default names are uniquified per parent
                                                      scope: Conv, Conv_1, ... (variable_scope.py)
    slim.conv2d  (layers.convolution)                 1x1 only: y = x . W[Cin,Cout] (+ biases unless a
                                                      normalizer_fn is given), then normalizer_fn(y, **params),
                                                      then activation_fn (DEFAULT tf.nn.relu); variables
                                                      <scope>/weights, <scope>/biases; weights_regularizer(W)
                                                      appended to REGULARIZATION_LOSSES
    slim.batch_norm (is_training default True)        tf.nn.moments over all but the last axis (biased variance),
                                                      tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma,
                                                      y = x*inv + (beta - mean*inv); moving averages updated
                                                      as mov -= (1-decay)*(mov - batch) into UPDATE_OPS
    slim.dropout -> tf.nn.dropout                     binary = floor(keep_prob + random_uniform(shape)),
                                                      y = x / keep_prob * binary; identity if not is_training
    slim.l2_regularizer(s)                            s * sum(w^2)/2; `None` for s == 0
    tf.nn.softmax                                     exp(x - max) / sum over the LAST axis
    tf.reduce_mean / reduce_sum / transpose / reshape / stack / concat / split / unstack / squeeze /
    expand_dims / where / equal / greater / less / to_float / square / ones / shape
    tf.image.resize_images (bilinear, TF 1.1 kernel)  scale = in/out as float32, in = i*scale (float32), lower =
                                                      floor(in), upper = min(ceil(in), size-1), lerp = in - lower
    tf.losses.softmax_cross_entropy / mean_squared_error / sigmoid_cross_entropy / add_loss
                                                      per-element losses reduced by compute_weighted_loss:
                                                      sum(losses*w) / #(elements with w != 0)
    tf.nn.weighted_cross_entropy_with_logits          (1-z)*x + (1+(q-1)*z) * (log1p(exp(-|x|)) + max(-x,0))
    tf.nn.sigmoid_cross_entropy_with_logits           max(x,0) - x*z + log1p(exp(-|x|))

Numbers are torch float64 tensors and torch autograd plays tf.gradients.  TF's float32 kernel rounding and
its RNG streams are NOT modelled (random draws come from a seeded numpy stream that the generator records, so
that the same dropout mask / uniform draws can be replayed through the oracle and the HIP path).
"""
from __future__ import annotations

import collections as collections_module
import contextlib
import functools
import math
import sys
import types
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch

DT = torch.float64


# ------------------------------------------------------------------------------------------- tensors
class Dimension(object):
    def __init__(self, v):
        self.value = v

    def __int__(self):
        return int(self.value)

    __index__ = __int__

    def __eq__(self, o):
        return int(self) == int(o)

    def __hash__(self):
        return hash(int(self))

    def __repr__(self):
        return 'Dimension(%d)' % self.value


class TensorShape(object):
    def __init__(self, dims):
        self._dims = [int(d) for d in dims]

    @property
    def ndims(self):
        return len(self._dims)

    def as_list(self):
        return list(self._dims)

    def num_elements(self):
        n = 1
        for d in self._dims:
            n *= d
        return n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return TensorShape(self._dims[i])
        return Dimension(self._dims[i])

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(Dimension(d) for d in self._dims)

    def __repr__(self):
        return 'TensorShape(%r)' % (self._dims,)


class Tensor(object):
    """A TF graph tensor, evaluated eagerly: `.v` is a torch tensor (float64, bool or int64)."""
    __array_priority__ = 1000

    def __init__(self, v, name=None):
        self.v = v
        self.name = name

    def get_shape(self):
        return TensorShape(self.v.shape)

    @property
    def shape(self):
        return TensorShape(self.v.shape)

    @property
    def dtype(self):
        return self.v.dtype

    def set_shape(self, shape):
        assert len(shape) == self.v.dim() and all(s is None or int(s) == d for s, d in zip(shape, self.v.shape))

    def __mul__(self, o):
        return Tensor(self.v * _raw(o, like=self.v))

    __rmul__ = __mul__

    def __add__(self, o):
        return Tensor(self.v + _raw(o, like=self.v))

    __radd__ = __add__

    def __sub__(self, o):
        return Tensor(self.v - _raw(o, like=self.v))

    def __rsub__(self, o):
        return Tensor(_raw(o, like=self.v) - self.v)

    def __truediv__(self, o):
        return Tensor(self.v / _raw(o, like=self.v))

    def __rtruediv__(self, o):
        return Tensor(_raw(o, like=self.v) / self.v)

    def __neg__(self):
        return Tensor(-self.v)

    def __getitem__(self, idx):
        return Tensor(self.v[idx])

    def __repr__(self):
        return 'Tensor(shape=%r, name=%r)' % (list(self.v.shape), self.name)


class _OpRef(object):
    def __init__(self, name):
        self.name = name


class Variable(Tensor):
    """tf.Variable: `.name` carries the ':0' output suffix, `.op.name` does not."""

    def __init__(self, v, full_name):
        Tensor.__init__(self, v, name=full_name + ':0')
        self.op = _OpRef(full_name)


def _raw(x, like=None):
    """python scalars / lists / numpy / Tensor -> torch tensor (dtype follows `like` for python numbers, as
    TF's op-def type inference does for `tf.where(v, loss_val, [0] * n)`)."""
    if isinstance(x, Tensor):
        return x.v
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, Dimension):
        x = int(x)
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], Tensor):
        return torch.stack([t.v for t in x])
    dt = like.dtype if (like is not None and like.dtype.is_floating_point) else None
    t = torch.as_tensor(np.asarray(x))
    if t.dtype.is_floating_point or dt is not None:
        t = t.to(dt or DT)
    return t


def _axes(axis, nd):
    if axis is None:
        return tuple(range(nd))
    if isinstance(axis, int):
        axis = [axis]
    return tuple(a % nd for a in axis)


# ------------------------------------------------------------------------------------ graph-level state
class Graph(object):
    """Everything a tf.Graph would hold for one build: variables by name, collections, the random source."""

    def __init__(self, value_fn: Callable[[str, List[int], Dict[str, Any]], np.ndarray],
                 uniform_fn: Callable[[List[int], str], np.ndarray]):
        self.variables: Dict[str, torch.Tensor] = {}
        self.var_init: Dict[str, Dict[str, Any]] = {}
        self.var_order: List[str] = []
        self.collections: Dict[str, list] = {}
        self.scope_stack: List[str] = []
        self.reuse_stack: List[bool] = []
        self.name_stack: List[str] = []                 # tf.name_scope (only used to tag LOSSES per clone)
        self.created: Dict[str, 'Variable'] = {}        # variables created in the CURRENT run of the graph
        self.default_name_counts: Dict[str, int] = {}     # variable_scopes_count of the variable store
        self.arg_stack: List[Dict[str, Dict[str, Any]]] = [{}]
        self.value_fn = value_fn
        self.uniform_fn = uniform_fn
        self.random_draws: List[Dict[str, Any]] = []
        self.log: List[str] = []

    def scope_name(self):
        return '/'.join(self.scope_stack)

    def add_to_collection(self, key, value):
        self.collections.setdefault(key, []).append(value)

    def get_collection(self, key, scope=None):
        items = list(self.collections.get(key, []))
        if scope:                                       # tf.get_collection(key, scope): re.match(scope, item.name)
            items = [i for i in items if (getattr(i, 'name', None) or '').startswith(scope)]
        return items

    def name_scope_prefix(self):
        return ''.join(n + '/' for n in self.name_stack if n)

    def begin_run(self):
        """A new evaluation of the same graph (one `session.run`): the variables keep their values, everything
        derived from them (collections, regularisation losses) is rebuilt by running the graph code again."""
        self.collections = {}
        self.created = {}
        self.default_name_counts = {}

    def get_variable(self, name, shape, initializer, regularizer=None, trainable=True):
        full = (self.scope_name() + '/' if self.scope_stack else '') + name
        reuse = any(self.reuse_stack)
        if full in self.created:
            if not reuse:
                raise ValueError('Variable %s already exists, disallowed. Did you mean to set reuse=True?' % full)
            return self.created[full]
        if reuse:
            raise ValueError('Variable %s does not exist, or was not created with tf.get_variable()' % full)
        shape = [int(s) for s in shape]
        if full in self.variables:                      # a later run of the same graph: the stored value
            t = self.variables[full]
            assert list(t.shape) == shape, full
        else:
            desc = dict(initializer or {'kind': 'unknown'})
            val = np.asarray(self.value_fn(full, shape, desc), dtype=np.float64).reshape(shape)
            t = torch.from_numpy(val.copy()).to(DT)
            if trainable:
                t.requires_grad_(True)
            self.variables[full] = t
            self.var_init[full] = desc
            self.var_order.append(full)
        var = Variable(t, full)
        self.created[full] = var
        if trainable:
            self.add_to_collection(GraphKeys.TRAINABLE_VARIABLES, var)
        if regularizer is not None:                     # applied once, where the variable is created
            loss = regularizer(Tensor(t))
            if loss is not None:
                self.add_to_collection(GraphKeys.REGULARIZATION_LOSSES, loss)
        return var


_GRAPH: Optional[Graph] = None


def set_graph(g: Optional[Graph]):
    global _GRAPH
    _GRAPH = g


def graph() -> Graph:
    assert _GRAPH is not None, 'tf1_shim: no Graph installed (set_graph)'
    return _GRAPH


class GraphKeys(object):
    LOSSES = 'losses'
    REGULARIZATION_LOSSES = 'regularization_losses'
    UPDATE_OPS = 'update_ops'
    TRAINABLE_VARIABLES = 'trainable_variables'


# ---------------------------------------------------------------------------------- scopes (tf / slim)
class _VarScope(object):
    def __init__(self, name):
        self.name = name
        self.original_name_scope = name + '/'            # what slim hands to collect_named_outputs


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, values=None, reuse=None):
    """tf.variable_scope, including the bookkeeping that decides the names of UNNAMED layers (variable_scope.py:
    _get_unique_variable_scope / open_variable_scope / close_variable_subscopes): every opened scope is counted
    under its full name; a default name takes the first suffix ('', '_1', '_2', ...) whose count is zero; leaving
    a scope resets the counts of all its sub-scopes -- which is why the second clone of a model, re-entering the
    same named scopes with reuse=True, finds 'Conv' again and not 'Conv_1'."""
    g = graph()
    counts = g.default_name_counts
    saved = list(g.scope_stack)
    if name_or_scope is None:
        parent = g.scope_name()
        full = lambda n: (parent + '/' + n) if parent else n
        name = default_name
        idx = 0
        while counts.get(full(name), 0) > 0:
            idx += 1
            name = '%s_%d' % (default_name, idx)
        g.scope_stack.append(name)
    elif isinstance(name_or_scope, _VarScope):           # a captured scope is re-entered by its ABSOLUTE name
        g.scope_stack[:] = [p for p in name_or_scope.name.split('/') if p]
    else:
        g.scope_stack.append(name_or_scope)
    opened = g.scope_name()
    counts[opened] = counts.get(opened, 0) + 1
    g.reuse_stack.append(bool(reuse))
    try:
        yield _VarScope(opened)
    finally:
        for k in list(counts):
            if not opened or k.startswith(opened + '/'):
                counts[k] = 0
        g.scope_stack[:] = saved
        g.reuse_stack.pop()


def get_variable_scope():
    return _VarScope(graph().scope_name())


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    """yields the scope's full name with the trailing '/', as TF does ('' for the empty scope)"""
    if _GRAPH is None:
        yield name
        return
    g = graph()
    g.name_stack.append(name or '')
    try:
        yield g.name_scope_prefix()
    finally:
        g.name_stack.pop()


def _op_key(op):
    return getattr(op, '_arg_scope_key', None) or (op.__module__ + '.' + op.__name__)


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    g = graph()
    if isinstance(list_ops_or_scope, dict):
        if kwargs:
            raise ValueError('When attempting to re-use a scope by suppling a dictionary, kwargs must be empty.')
        g.arg_stack.append({k: dict(v) for k, v in list_ops_or_scope.items()})
        try:
            yield list_ops_or_scope
        finally:
            g.arg_stack.pop()
        return
    if not isinstance(list_ops_or_scope, (list, tuple)):
        raise TypeError('list_ops_or_scope must either be a list/tuple or reused scope (i.e. dict)')
    cur = {k: dict(v) for k, v in g.arg_stack[-1].items()}
    for op in list_ops_or_scope:
        key = _op_key(op)
        if not getattr(op, '_has_arg_scope', False):
            raise ValueError('%s is not decorated with @add_arg_scope' % key)
        merged = dict(cur.get(key, {}))
        merged.update(kwargs)
        cur[key] = merged
    g.arg_stack.append(cur)
    try:
        yield cur
    finally:
        g.arg_stack.pop()


def add_arg_scope(func):
    key = func.__module__ + '.' + func.__name__

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        defaults = graph().arg_stack[-1].get(key, {}) if _GRAPH is not None else {}
        if defaults:
            merged = dict(defaults)
            merged.update(kwargs)
            kwargs = merged
        return func(*args, **kwargs)

    wrapper._arg_scope_key = key
    wrapper._has_arg_scope = True
    return wrapper


# ------------------------------------------------------------------------------------------ initializers
def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    return {'kind': 'random_normal', 'mean': float(mean), 'stddev': float(stddev)}


def zeros_initializer(dtype=None):
    return {'kind': 'zeros'}


def ones_initializer(dtype=None):
    return {'kind': 'ones'}


def constant_initializer(value=0, dtype=None):
    return {'kind': 'constant', 'value': float(value)}


def variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False, seed=None, dtype=None):
    # tf.contrib.layers: truncated normal with stddev = sqrt(1.3 * factor / n), n = fan_in for FAN_IN
    return {'kind': 'variance_scaling', 'factor': float(factor), 'mode': mode, 'uniform': bool(uniform),
            'truncated_normal_stddev': 'sqrt(1.3*factor/n)'}


def xavier_initializer(uniform=True, seed=None, dtype=None):
    return {'kind': 'xavier', 'uniform': bool(uniform)}


# ------------------------------------------------------------------------------------------------ slim ops
def l2_regularizer(scale, scope=None):
    if isinstance(scale, (int, float)) and scale == 0.:
        return lambda _: None

    def l2(weights):
        return Tensor(float(scale) * (weights.v ** 2).sum() / 2.0)       # scale * tf.nn.l2_loss(w)
    return l2


@add_arg_scope
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None,
               param_initializers=None, param_regularizers=None, updates_collections=GraphKeys.UPDATE_OPS,
               is_training=True, reuse=None, variables_collections=None, outputs_collections=None,
               trainable=True, batch_weights=None, fused=False, data_format='NHWC',
               zero_debias_moving_mean=False, scope=None):
    g = graph()
    x = inputs.v
    c = x.shape[-1]
    with variable_scope(scope, 'BatchNorm', [inputs], reuse=reuse):
        beta = g.get_variable('beta', [c], zeros_initializer(), trainable=trainable).v if center else None
        gamma = g.get_variable('gamma', [c], ones_initializer(), trainable=trainable).v if scale else None
        moving_mean = g.get_variable('moving_mean', [c], zeros_initializer(), trainable=False).v
        moving_variance = g.get_variable('moving_variance', [c], ones_initializer(), trainable=False).v
        if is_training:
            axes = tuple(range(x.dim() - 1))
            mean = x.mean(dim=axes)                                            # tf.nn.moments
            variance = ((x - mean) ** 2).mean(dim=axes)
            g.add_to_collection(updates_collections or GraphKeys.UPDATE_OPS,
                                ('moving_mean', (moving_mean - (1 - decay) * (moving_mean - mean)).detach()))
            g.add_to_collection(updates_collections or GraphKeys.UPDATE_OPS,
                                ('moving_variance',
                                 (moving_variance - (1 - decay) * (moving_variance - variance)).detach()))
        else:
            mean, variance = moving_mean, moving_variance
        inv = torch.rsqrt(variance + epsilon)                                   # tf.nn.batch_normalization
        if gamma is not None:
            inv = inv * gamma
        y = x * inv + ((beta - mean * inv) if beta is not None else (-mean * inv))
    out = Tensor(y)
    if activation_fn is not None:
        out = activation_fn(out)
    return out


@add_arg_scope
def conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', data_format=None, rate=1,
           activation_fn='__default_relu__', normalizer_fn=None, normalizer_params=None,
           weights_initializer=None, weights_regularizer=None, biases_initializer='__default_zeros__',
           biases_regularizer=None, reuse=None, variables_collections=None, outputs_collections=None,
           trainable=True, scope=None):
    g = graph()
    if activation_fn == '__default_relu__':
        activation_fn = relu                              # layers.convolution: activation_fn=nn.relu
    if biases_initializer == '__default_zeros__':
        biases_initializer = zeros_initializer()
    if weights_initializer is None:
        weights_initializer = xavier_initializer()
    ks = [kernel_size, kernel_size] if isinstance(kernel_size, int) else list(kernel_size)
    x = inputs.v
    cin = x.shape[-1]
    cout = int(num_outputs)
    general = ks != [1, 1] or stride != 1 or rate != 1
    with variable_scope(scope, 'Conv', [inputs], reuse=reuse) as sc:
        w = g.get_variable('weights', ks + [cin, cout],
                           dict(weights_initializer, fan_in=cin * ks[0] * ks[1], fan_out=cout * ks[0] * ks[1])
                           if general else dict(weights_initializer, fan_in=cin, fan_out=cout),
                           regularizer=weights_regularizer, trainable=trainable)
        if general:
            y = _conv2d_nhwc(x, w.v, int(stride), int(rate), padding)
        else:
            y = torch.matmul(x, w.v.reshape(cin, cout))
        if normalizer_fn is None and biases_initializer is not None:
            b = g.get_variable('biases', [cout], biases_initializer, regularizer=biases_regularizer,
                               trainable=trainable)
            y = y + b.v                                    # nn.bias_add
        out = Tensor(y)
        if normalizer_fn is not None:
            out = normalizer_fn(out, **(normalizer_params or {}))
        if activation_fn is not None:
            out = activation_fn(out)
    return collect_named_outputs(outputs_collections, sc.original_name_scope, out)


def _same_pad(size, k_eff, s):
    """TF 'SAME': out = ceil(size / s); total = max((out - 1) s + k_eff - size, 0), split low = total // 2"""
    out = -(-size // s)
    total = max((out - 1) * s + k_eff - size, 0)
    return total // 2, total - total // 2


def _conv2d_nhwc(x, w_hwio, stride, rate, padding):
    """tf.nn.convolution on NHWC / HWIO (Conv2D; atrous via dilation), 'SAME' or 'VALID'"""
    assert x.dim() == 4
    kh, kw = int(w_hwio.shape[0]), int(w_hwio.shape[1])
    xn = x.permute(0, 3, 1, 2)
    if padding == 'SAME':
        pt, pb = _same_pad(xn.shape[2], (kh - 1) * rate + 1, stride)
        pl, pr = _same_pad(xn.shape[3], (kw - 1) * rate + 1, stride)
        xn = torch.nn.functional.pad(xn, (pl, pr, pt, pb))
    elif padding != 'VALID':
        raise ValueError(padding)
    y = torch.nn.functional.conv2d(xn, w_hwio.permute(3, 2, 0, 1), stride=stride, dilation=rate)
    return y.permute(0, 2, 3, 1)


def collect_named_outputs(collections, alias, outputs):
    """slim.utils.collect_named_outputs: tags the tensor with `alias` (trailing '/' dropped) and files it"""
    if collections:
        if alias[-1] == '/':
            alias = alias[:-1]
        outputs.alias = alias
        for c in ([collections] if isinstance(collections, str) else collections):
            graph().add_to_collection(c, outputs)
    return outputs


def convert_collection_to_dict(collection):
    return collections_module.OrderedDict((t.alias, t) for t in graph().get_collection(collection))


def last_dimension(shape, min_rank=1):
    dims = shape.as_list()
    if len(dims) < min_rank:
        raise ValueError('rank of shape must be at least %d not: %d' % (min_rank, len(dims)))
    return dims[-1]


def pad(tensor, paddings, mode='CONSTANT', name=None):
    assert mode == 'CONSTANT'
    v = _raw(tensor)
    flat = []
    for lo, hi in reversed([[int(a), int(b)] for a, b in paddings]):
        flat += [lo, hi]
    return Tensor(torch.nn.functional.pad(v, flat))


@add_arg_scope
def dropout(inputs, keep_prob=0.5, noise_shape=None, is_training=True, outputs_collections=None, scope=None):
    if not is_training:
        return inputs                                      # utils.smart_cond -> identity
    return nn_dropout(inputs, keep_prob)


@add_arg_scope
def max_pool2d(inputs, kernel_size, stride=2, padding='VALID', data_format='NHWC', outputs_collections=None,
               scope=None):
    """slim.max_pool2d (nn.max_pool): 'SAME' padding never wins the maximum"""
    ks = [kernel_size, kernel_size] if isinstance(kernel_size, int) else list(kernel_size)
    st = [stride, stride] if isinstance(stride, int) else list(stride)
    xn = _raw(inputs).permute(0, 3, 1, 2)
    if padding == 'SAME':
        pt, pb = _same_pad(xn.shape[2], ks[0], st[0])
        pl, pr = _same_pad(xn.shape[3], ks[1], st[1])
        xn = torch.nn.functional.pad(xn, (pl, pr, pt, pb), value=float('-inf'))
    elif padding != 'VALID':
        raise ValueError(padding)
    y = torch.nn.functional.max_pool2d(xn, ks, st).permute(0, 2, 3, 1)
    with variable_scope(scope, 'MaxPool2D', [inputs]) as sc:
        return collect_named_outputs(outputs_collections, sc.original_name_scope, Tensor(y))


@add_arg_scope
def avg_pool2d(*a, **k):
    raise NotImplementedError


@add_arg_scope
def fully_connected(*a, **k):
    raise NotImplementedError


def one_hot_encoding(labels, num_classes, on_value=1.0, off_value=0.0, outputs_collections=None, scope=None):
    lab = _raw(labels).long()
    out = torch.full((lab.shape[0], int(num_classes)), float(off_value), dtype=DT)
    out[torch.arange(lab.shape[0]), lab] = float(on_value)
    return Tensor(out)


# --------------------------------------------------------------------------------------------------- tf.nn
def relu(x, name=None):
    return Tensor(torch.clamp(_raw(x), min=0))


def softmax(logits, dim=-1, name=None):
    x = _raw(logits)
    e = torch.exp(x - x.max(dim=dim, keepdim=True).values)
    return Tensor(e / e.sum(dim=dim, keepdim=True))


def nn_dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    g = graph()
    xv = _raw(x)
    u = g.uniform_fn(list(xv.shape), 'dropout')
    g.random_draws.append({'kind': 'dropout', 'keep_prob': float(keep_prob), 'uniform': u})
    random_tensor = float(keep_prob) + torch.from_numpy(np.asarray(u, dtype=np.float64))
    binary_tensor = torch.floor(random_tensor)
    return Tensor(xv / float(keep_prob) * binary_tensor)        # math_ops.div(x, keep_prob) * binary_tensor


def sigmoid_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, name=None):
    x, z = _raw(logits), _raw(labels).to(DT)
    return Tensor(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs())))


def weighted_cross_entropy_with_logits(targets, logits, pos_weight, name=None):
    x, z = _raw(logits), _raw(targets).to(DT)
    log_weight = 1 + (float(pos_weight) - 1) * z
    return Tensor((1 - z) * x + log_weight * (torch.log1p(torch.exp(-x.abs())) + torch.clamp(-x, min=0)))


def softmax_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, dim=-1, name=None):
    x, z = _raw(logits), _raw(labels).to(DT)
    sh = x - x.max(dim=-1, keepdim=True).values
    lsm = sh - torch.log(torch.exp(sh).sum(dim=-1, keepdim=True))
    return Tensor(-(z * lsm).sum(dim=-1))


# ------------------------------------------------------------------------------------------- tf.* array ops
def reshape(t, shape, name=None):
    return Tensor(_raw(t).reshape([int(s) for s in shape]))


def transpose(t, perm=None, name=None):
    v = _raw(t)
    if perm is None:
        perm = list(range(v.dim()))[::-1]
    if len(perm) != v.dim():
        raise ValueError('transpose: perm %r does not match a rank-%d tensor' % (perm, v.dim()))
    return Tensor(v.permute(*perm))


def stack(values, axis=0, name=None):
    return Tensor(torch.stack([_raw(v) for v in values], dim=axis))


def unstack(value, num=None, axis=0, name=None):
    return [Tensor(t) for t in torch.unbind(_raw(value), dim=axis)]


def concat(values, axis, name=None):
    return Tensor(torch.cat([_raw(v) for v in values], dim=axis))


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    v = _raw(value)
    if isinstance(num_or_size_splits, (list, tuple)):
        return [Tensor(t) for t in torch.split(v, [int(s) for s in num_or_size_splits], dim=axis)]
    n = int(num_or_size_splits)
    if v.shape[axis] % n:
        raise ValueError('split: dimension %d not divisible by %d' % (v.shape[axis], n))
    return [Tensor(t) for t in torch.split(v, v.shape[axis] // n, dim=axis)]


def squeeze(t, axis=None, name=None, squeeze_dims=None):
    v = _raw(t)
    axis = squeeze_dims if axis is None else axis
    if axis is None:
        return Tensor(v.squeeze())
    for a in sorted(_axes(axis, v.dim()), reverse=True):
        if v.shape[a] != 1:
            raise ValueError('squeeze: dimension %d is %d, not 1' % (a, v.shape[a]))
        v = v.squeeze(a)
    return Tensor(v)


def expand_dims(t, axis=None, name=None, dim=None):
    v = _raw(t)
    axis = dim if axis is None else axis
    if axis < 0:
        axis = v.dim() + 1 + axis
    return Tensor(v.unsqueeze(axis))


def reduce_mean(t, axis=None, keep_dims=False, name=None, reduction_indices=None):
    v = _raw(t)
    ax = _axes(reduction_indices if axis is None else axis, v.dim())
    return Tensor(v.sum(dim=ax, keepdim=keep_dims) / float(np.prod([v.shape[a] for a in ax])))


def reduce_sum(t, axis=None, keep_dims=False, name=None, reduction_indices=None):
    if isinstance(t, (list, tuple)):            # a python list of scalars is packed into a vector first
        v = torch.stack([_raw(e).to(DT).reshape(()) for e in t])
    else:
        v = _raw(t)
    ax = _axes(reduction_indices if axis is None else axis, v.dim())
    return Tensor(v.sum(dim=ax, keepdim=keep_dims))


def where(condition, x=None, y=None, name=None):
    c = _raw(condition).bool()
    xv = _raw(x)
    yv = _raw(y, like=xv)
    if c.dim() == 1 and xv.dim() > 1:
        c = c.reshape([-1] + [1] * (xv.dim() - 1))
    return Tensor(torch.where(c, xv, yv))


def equal(a, b, name=None):
    av = _raw(a)
    return Tensor(av == _raw(b, like=av))


def greater(a, b, name=None):
    av = _raw(a)
    return Tensor(av > _raw(b, like=av))


def less(a, b, name=None):
    av = _raw(a)
    return Tensor(av < _raw(b, like=av))


def to_float(t, name=None):
    return Tensor(_raw(t).to(DT))


def square(t, name=None):
    v = _raw(t)
    return Tensor(v * v)


def shape(t, name=None):
    return list(_raw(t).shape)


def ones(shape_, dtype=None, name=None):
    return Tensor(torch.ones([int(s) for s in shape_], dtype=DT))


def zeros(shape_, dtype=None, name=None):
    return Tensor(torch.zeros([int(s) for s in shape_], dtype=DT))


def random_uniform(shape_, minval=0, maxval=None, dtype=None, seed=None, name=None):
    g = graph()
    maxval = 1.0 if maxval is None else maxval
    u = np.asarray(g.uniform_fn([int(s) for s in shape_], 'random_uniform'), dtype=np.float64)
    g.random_draws.append({'kind': 'random_uniform', 'uniform': u})
    return Tensor(torch.from_numpy(u * (maxval - minval) + minval))


def argmax(t, axis=None, name=None, dimension=None):
    return Tensor(torch.argmax(_raw(t), dim=dimension if axis is None else axis))


def identity(t, name=None):
    return t


# ---------------------------------------------------------------------------------------------- tf.image
def resize_images(images, size, method=0, align_corners=False):
    """ResizeBilinear as the TF 1.1 CPU kernel computes it (core/kernels/resize_bilinear_op.cc +
    image_resizer_state.h, align_corners=False): scale and source coordinates in FLOAT32."""
    if method != 0 or align_corners:
        raise NotImplementedError
    v = _raw(images)
    squeeze0 = v.dim() == 3
    if squeeze0:
        v = v.unsqueeze(0)
    n, h, w, c = v.shape
    oh, ow = int(size[0]), int(size[1])
    if (oh, ow) == (h, w):
        return images                                   # resize_images returns the input unchanged

    def weights(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        lo, hi, lerp = [], [], []
        for i in range(out_size):
            src = np.float32(i) * scale                 # float32 product
            lo.append(int(np.floor(src)))
            hi.append(min(int(np.ceil(src)), in_size - 1))
            lerp.append(float(np.float32(src - np.floor(src))))
        return lo, hi, torch.tensor(lerp, dtype=DT)

    y0, y1, fy = weights(oh, h)
    x0, x1, fx = weights(ow, w)
    fy = fy.reshape(1, oh, 1, 1)
    fx = fx.reshape(1, 1, ow, 1)
    tl, tr = v[:, y0][:, :, x0], v[:, y0][:, :, x1]
    bl, br = v[:, y1][:, :, x0], v[:, y1][:, :, x1]
    top = tl + (tr - tl) * fx                            # compute_lerp
    bot = bl + (br - bl) * fx
    out = top + (bot - top) * fy
    return Tensor(out[0] if squeeze0 else out)


# ---------------------------------------------------------------------------------------------- label path ops
# (src/preprocess_pipeline.py: _replay_augmentation and the normalise / resize lines; evaluated in FLOAT32
#  -- set_float_dtype(torch.float32) -- because that graph is float32 and its truncations depend on it)
def set_float_dtype(dt):
    """The float type of every tensor the stand-in creates (float64 for the head graphs, float32 for the label
    pipeline)."""
    global DT
    DT = dt


_DTYPES = {'float32': None, 'uint8': torch.uint8, 'int32': torch.int32, 'int64': torch.int64, 'bool': torch.bool}


def cast(t, dtype, name=None):
    """tf.cast: float -> integer truncates toward zero; 'float32' maps to the stand-in's float type."""
    v = _raw(t)
    if dtype not in _DTYPES:
        raise NotImplementedError('cast to %r' % (dtype,))
    return Tensor(v.to(_DTYPES[dtype] or DT))


def to_int32(t, name=None):
    return cast(t, 'int32')


def constant(value, dtype=None, name=None):
    return Tensor(_raw(value))


def tf_tuple(tensors, name=None):
    return list(tensors)


def cond(pred, true_fn=None, false_fn=None, name=None, fn1=None, fn2=None):
    p = bool(_raw(pred).item()) if not isinstance(pred, bool) else pred
    return (true_fn or fn1)() if p else (false_fn or fn2)()


def tf_slice(input_, begin, size, name=None):
    """tf.slice: begin/size per dimension (size -1 = to the end); out-of-range requests are an error."""
    v = _raw(input_)
    b = [int(x) for x in _raw(begin).tolist()]
    s = [int(x) for x in _raw(size).tolist()]
    if len(b) != v.dim() or len(s) != v.dim():
        raise ValueError('slice: begin/size of length %d/%d for a rank-%d tensor' % (len(b), len(s), v.dim()))
    idx = []
    for d, (bi, si) in enumerate(zip(b, s)):
        if si == -1:
            si = v.shape[d] - bi
        if bi < 0 or si < 0 or bi + si > v.shape[d]:
            raise ValueError('slice: [%d, %d) outside dimension %d of size %d' % (bi, bi + si, d, v.shape[d]))
        idx.append(slice(bi, bi + si))
    return Tensor(v[tuple(idx)])


def reduce_min(t, axis=None, keep_dims=False, name=None):
    v = _raw(t)
    return Tensor(v.amin(dim=_axes(axis, v.dim()), keepdim=keep_dims))


def reduce_max(t, axis=None, keep_dims=False, name=None):
    v = _raw(t)
    return Tensor(v.amax(dim=_axes(axis, v.dim()), keepdim=keep_dims))


def flip_left_right(image):
    """tf.image.flip_left_right: reverse the width axis of a [height, width, channels] image."""
    v = _raw(image)
    if v.dim() != 3:
        raise ValueError('flip_left_right: rank-%d input' % v.dim())
    return Tensor(torch.flip(v, dims=[1]))


def convert_image_dtype(image, dtype, saturate=False, name=None):
    """tf.image.convert_image_dtype, integer -> float: cast, then MULTIPLY by 1 / dtype.max (image_ops_impl.py)."""
    v = _raw(image)
    if dtype != 'float32':
        raise NotImplementedError
    if v.dtype.is_floating_point:
        return Tensor(v.to(DT))
    if v.dtype != torch.uint8:
        raise NotImplementedError
    return Tensor(v.to(DT) * torch.tensor(1.0 / 255, dtype=DT))


# ---------------------------------------------------------------------------------------- training graph
# src/train.py builds ONE graph and then calls session.run on different fetches (train_ops[i]); what a fetch
# evaluates depends on graph dependencies.  The model part of the graph is evaluated eagerly here, so a
# `session.run` is modelled as: dequeue fresh batches, run the (reference) model-building code again on them
# with the variables' current values (Graph.begin_run), then execute the fetched ops.  Ops and tensors that
# are built ONCE by the reference's training code (assign / assign_add / group / control_dependencies /
# apply_gradients / with_dependencies) are lazy nodes evaluated per run, each at most once per run.
class Run(object):
    """one session.run: memo of what has been evaluated"""

    def __init__(self, session):
        self.session = session
        self.done: Dict[int, Any] = {}


class Node(object):
    """A lazy graph node: `deps` run first (control inputs), then `fn(run)`; memoised per run."""

    def __init__(self, fn, deps=(), name=None, shape=None):
        self.fn = fn
        self.deps = list(deps) + list(_CONTROL_DEPS[-1]) if _CONTROL_DEPS else list(deps)
        self.name = name
        self._shape = shape

    def eval(self, run):
        k = id(self)
        if k not in run.done:
            for d in self.deps:
                d.eval(run)
            run.done[k] = self.fn(run)
        return run.done[k]

    # the tensor surface the reference's accumulation block touches
    def get_shape(self):
        return TensorShape(self._shape)

    def __truediv__(self, o):
        return Node(lambda run: self.eval(run) / _lazy_value(o, run), shape=self._shape)


_CONTROL_DEPS: List[List[Node]] = []


def _lazy_value(x, run):
    if isinstance(x, Node):
        return x.eval(run)
    if isinstance(x, Tensor):
        return x.v
    return x


@contextlib.contextmanager
def control_dependencies(ops):
    _CONTROL_DEPS.append((list(_CONTROL_DEPS[-1]) if _CONTROL_DEPS else []) + list(ops))
    try:
        yield
    finally:
        _CONTROL_DEPS.pop()


def group(*ops, **kw):
    return Node(lambda run: None, deps=ops, name=kw.get('name'))


def with_dependencies(dependencies, output_tensor, name=None):
    return Node(lambda run: _lazy_value(output_tensor, run), deps=dependencies, name=name)


class LocalVariable(Node):
    """slim.local_variable(initial_value, name=...): a non-trainable variable; reading it is a lazy node."""

    def __init__(self, initial_value, name=None):
        self.value = torch.as_tensor(np.asarray(initial_value)).to(DT).clone()
        Node.__init__(self, lambda run: self.value, name=name, shape=list(self.value.shape))
        self.deps = []                                    # the read op is created with the variable

    def eval(self, run):                                  # never memoised: a read sees the latest assignment
        return self.value

    def assign(self, t):
        def fn(run):
            self.value = _lazy_value(t, run).detach().clone()
        return Node(fn, name='assign')

    def assign_add(self, t):
        def fn(run):
            self.value = self.value + _lazy_value(t, run).detach()
        return Node(fn, name='assign_add')


def local_variable(initial_value, name=None, **kw):
    return LocalVariable(initial_value, name=(graph().scope_name() + '/' if graph().scope_stack else '') + (name or ''))


class GlobalStep(LocalVariable):
    def __init__(self):
        LocalVariable.__init__(self, np.zeros((), dtype=np.int64), name='global_step')
        self.value = torch.zeros((), dtype=torch.int64)


def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
    """tf.train.exponential_decay: lr * rate ** (step / decay_steps), the exponent floored when staircase
    (learning_rate_decay.py; computed in float32 there, in the stand-in's float type here)."""
    def fn(run):
        p = _lazy_value(global_step, run).to(DT) / float(decay_steps)
        if staircase:
            p = torch.floor(p)
        return torch.tensor(float(learning_rate), dtype=DT) * torch.tensor(float(decay_rate), dtype=DT) ** p
    return Node(fn, name=name, shape=[])


def polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False,
                     name=None):
    """tf.train.polynomial_decay, cycle=False: step = min(step, decay_steps);
    (lr - end) * (1 - step / decay_steps) ** power + end."""
    assert not cycle

    def fn(run):
        step = torch.clamp(_lazy_value(global_step, run).to(DT), max=float(decay_steps))
        return (float(learning_rate) - float(end_learning_rate)) * (1.0 - step / float(decay_steps)) ** float(power) \
            + float(end_learning_rate)
    return Node(fn, name=name, shape=[])


class MomentumOptimizer(object):
    """tf.train.MomentumOptimizer (training_ops ApplyMomentum, use_nesterov=False):
         accum = momentum * accum + grad;  var -= lr * accum;   then global_step += 1."""

    def __init__(self, learning_rate, momentum, use_locking=False, name='Momentum', use_nesterov=False):
        assert not use_nesterov
        self.learning_rate, self.momentum, self.name = learning_rate, float(momentum), name
        self.slots: Dict[str, torch.Tensor] = {}
        self.applied = 0

    def compute_gradients(self, loss, var_list=None, **kw):
        if kw:
            raise NotImplementedError(sorted(kw))
        vs = list(var_list) if var_list is not None else graph().get_collection(GraphKeys.TRAINABLE_VARIABLES)
        gs = torch.autograd.grad(_raw(loss), [v.v for v in vs], retain_graph=True, allow_unused=True)
        return [(None if g_ is None else Tensor(g_), v) for g_, v in zip(gs, vs)]

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gvs = list(grads_and_vars)

        def fn(run):
            lr = _lazy_value(self.learning_rate, run)
            with torch.no_grad():
                for g_, v in gvs:
                    if g_ is None:
                        continue
                    gval = _lazy_value(g_, run).detach()
                    acc = self.slots.get(v.op.name)
                    acc = gval.clone() if acc is None else self.momentum * acc + gval     # slot starts at zero
                    self.slots[v.op.name] = acc
                    v.v -= lr * acc
                if global_step is not None:
                    global_step.value = global_step.value + 1
            self.applied += 1
        return Node(fn, name=name or self.name)


class GradientDescentOptimizer(MomentumOptimizer):
    def __init__(self, learning_rate, use_locking=False, name='GradientDescent'):
        MomentumOptimizer.__init__(self, learning_rate, 0.0, name=name)


class Session(object):
    """`before_run()` is the per-run rebuild of the eager part of the graph (fresh batches)."""

    def __init__(self, before_run: Callable[[], None]):
        self.before_run = before_run
        self.runs = 0

    def run(self, fetches, options=None, run_metadata=None, feed_dict=None):
        self.before_run()
        self.runs += 1
        run = Run(self)
        single = not isinstance(fetches, (list, tuple))
        out = []
        for f in ([fetches] if single else fetches):
            val = _lazy_value(f, run)
            out.append(val.detach().numpy().copy() if isinstance(val, torch.Tensor) else val)
        return out[0] if single else out


def add_n(inputs, name=None):
    vs = [_raw(i) for i in inputs]
    out = vs[0]
    for v in vs[1:]:
        out = out + v
    return Tensor(out)


def div(x, y, name=None):
    xv = _raw(x)
    return Tensor(xv / _raw(y, like=xv))


@contextlib.contextmanager
def device(device_name_or_function=None):
    yield


def trainable_variables():
    return graph().get_collection(GraphKeys.TRAINABLE_VARIABLES)


@add_arg_scope
def model_variable(*a, **k):
    raise NotImplementedError


@add_arg_scope
def variable(*a, **k):
    raise NotImplementedError


# ---------------------------------------------------------------------------------------------- tf.losses
def add_loss(loss, loss_collection=GraphKeys.LOSSES):
    if loss_collection:
        t = loss if isinstance(loss, Tensor) else Tensor(_raw(loss).to(DT))
        if t.name is None:                                # tagged with the enclosing name scope ('clone_1/...'),
            t.name = graph().name_scope_prefix() + 'loss'  # which is what tf.get_collection(LOSSES, scope) filters on
        graph().add_to_collection(loss_collection, t)


def compute_weighted_loss(losses, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES):
    lv = _raw(losses).to(DT)
    wv = _raw(weights).to(DT) if not isinstance(weights, (int, float)) else torch.tensor(float(weights), dtype=DT)
    wb = torch.broadcast_to(wv, lv.shape) if wv.dim() == 0 or wv.shape == lv.shape else \
        torch.broadcast_to(wv.reshape(list(wv.shape) + [1] * (lv.dim() - wv.dim())), lv.shape)
    total_loss = (lv * wb).sum()                                           # _scale_losses
    num_present = (wb != 0).to(DT).sum()                                   # _num_present
    mean_loss = torch.where(num_present > 0, total_loss / torch.clamp(num_present, min=1.0),
                            torch.zeros_like(total_loss))                  # _safe_mean
    out = Tensor(mean_loss)
    add_loss(out, loss_collection)
    return out


def softmax_cross_entropy(onehot_labels, logits, weights=1.0, label_smoothing=0, scope=None,
                          loss_collection=GraphKeys.LOSSES):
    assert label_smoothing == 0
    losses = softmax_cross_entropy_with_logits(labels=onehot_labels, logits=logits)
    return compute_weighted_loss(losses, weights, scope, loss_collection)


def mean_squared_error(labels, predictions, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES):
    d = _raw(predictions) - _raw(labels).to(DT)
    return compute_weighted_loss(Tensor(d * d), weights, scope, loss_collection)


def sigmoid_cross_entropy(multi_class_labels, logits, weights=1.0, label_smoothing=0, scope=None,
                          loss_collection=GraphKeys.LOSSES):
    assert label_smoothing == 0
    losses = sigmoid_cross_entropy_with_logits(labels=to_float(multi_class_labels), logits=logits)
    return compute_weighted_loss(losses, weights, scope, loss_collection)


def get_losses(scope=None, loss_collection=GraphKeys.LOSSES):
    return graph().get_collection(loss_collection)


def get_regularization_losses(scope=None):
    return graph().get_collection(GraphKeys.REGULARIZATION_LOSSES)


# ------------------------------------------------------------------------------- module objects to install
def build_modules() -> Dict[str, types.ModuleType]:
    """Returns {'tensorflow': module, ...} ready for sys.modules."""
    tf = types.ModuleType('tensorflow')
    for fn in (reshape, transpose, stack, unstack, concat, split, squeeze, expand_dims, reduce_mean, reduce_sum,
               where, equal, greater, less, to_float, square, shape, ones, zeros, random_uniform, argmax,
               identity, variable_scope, name_scope, random_normal_initializer, zeros_initializer,
               ones_initializer, constant_initializer):
        setattr(tf, fn.__name__, fn)
    for fn in (cast, to_int32, constant, cond, reduce_min, reduce_max):
        setattr(tf, fn.__name__, fn)
    tf.slice, tf.tuple = tf_slice, tf_tuple
    for fn in (add_n, div, device, trainable_variables, get_variable_scope, control_dependencies, group):
        setattr(tf, fn.__name__, fn)
    train = types.ModuleType('tensorflow.train')
    train.exponential_decay = exponential_decay
    train.polynomial_decay = polynomial_decay
    train.MomentumOptimizer = MomentumOptimizer
    train.GradientDescentOptimizer = GradientDescentOptimizer
    tf.train = train
    summary = types.ModuleType('tensorflow.summary')
    summary.scalar = lambda *a, **k: None
    summary.histogram = lambda *a, **k: None
    tf.summary = summary
    tf.IndexedSlices = type('IndexedSlices', (), {})
    tf.NodeDef = type('NodeDef', (), {})
    tf.GraphKeys = GraphKeys
    tf.float32, tf.uint8, tf.int32, tf.int64, tf.bool = 'float32', 'uint8', 'int32', 'int64', 'bool'
    tf.Tensor = Tensor
    nn = types.ModuleType('tensorflow.nn')
    nn.relu, nn.softmax, nn.dropout = relu, softmax, nn_dropout
    nn.sigmoid_cross_entropy_with_logits = sigmoid_cross_entropy_with_logits
    nn.weighted_cross_entropy_with_logits = weighted_cross_entropy_with_logits
    nn.softmax_cross_entropy_with_logits = softmax_cross_entropy_with_logits
    tf.nn = nn
    image = types.ModuleType('tensorflow.image')
    image.resize_images = resize_images
    image.flip_left_right = flip_left_right
    image.convert_image_dtype = convert_image_dtype
    tf.image = image
    losses = types.ModuleType('tensorflow.losses')
    for fn in (add_loss, compute_weighted_loss, softmax_cross_entropy, mean_squared_error, sigmoid_cross_entropy,
               get_losses, get_regularization_losses):
        setattr(losses, fn.__name__, fn)
    tf.losses = losses
    logging = types.ModuleType('tensorflow.logging')
    logging.info = lambda *a, **k: graph().log.append(str(a[0]) if a else '') if _GRAPH is not None else None
    tf.logging = logging
    tf.get_collection = lambda key, scope=None: graph().get_collection(key, scope)
    tf.add_to_collection = lambda key, v: graph().add_to_collection(key, v)

    slim = types.ModuleType('tensorflow.contrib.slim')
    for fn in (arg_scope, add_arg_scope, conv2d, batch_norm, dropout, max_pool2d, avg_pool2d, fully_connected,
               l2_regularizer, variance_scaling_initializer, xavier_initializer, one_hot_encoding):
        setattr(slim, fn.__name__, fn)
    slim.losses = losses
    slim.model_variable, slim.variable, slim.local_variable = model_variable, variable, local_variable
    slim.softmax = lambda logits, scope=None: softmax(logits)      # default argument of the backbone builders
    utils = types.ModuleType('tensorflow.contrib.slim.utils')
    utils.collect_named_outputs = collect_named_outputs
    utils.convert_collection_to_dict = convert_collection_to_dict
    utils.last_dimension = last_dimension
    tf.pad = pad
    slim.utils = utils
    contrib = types.ModuleType('tensorflow.contrib')
    contrib.slim = slim
    tf.contrib = contrib
    return {'tensorflow': tf, 'tensorflow.contrib': contrib, 'tensorflow.contrib.slim': slim,
            'tensorflow.nn': nn, 'tensorflow.losses': losses}
