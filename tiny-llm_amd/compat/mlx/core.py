"""`mlx.core` facade over PyTorch for the symbols the reference's Week 1-3 tests and benches touch
(SURVEY.md §8b "Tensor facade needed"; semantics of the mx.* oracles: SURVEY.md Appendix A).

Arrays are plain ``torch.Tensor`` objects: the product (tiny_llm_hip) consumes and returns tensors, so nothing is wrapped.
Importing this module adds the MLX method names the tests call (`astype`, `.at[idx].add`, numpy-style `transpose(*perm)`,
`__array__` for device tensors) to ``torch.Tensor``.  New arrays are created on the default device: the GPU when one is
visible (MLX's unified memory has no placement; the course extension is GPU-only), else the CPU; ``mx.stream(mx.cpu)``
switches the default for its block.
"""

from __future__ import annotations

import contextlib
import math as _math
from types import SimpleNamespace as _NS

import numpy as _np
import torch as _torch

__version__ = "0.32.0+torch-facade"

# ---- dtypes ---------------------------------------------------------------------------------------------------------
Dtype = _torch.dtype
float32, float16, bfloat16 = _torch.float32, _torch.float16, _torch.bfloat16
int8, int16, int32, int64, uint8, bool_ = _torch.int8, _torch.int16, _torch.int32, _torch.int64, _torch.uint8, _torch.bool
# Packed W4 words travel as int32 bit containers in the product (torch has no general uint32 arithmetic)
uint32 = _torch.int32
inf = float("inf")
pi = _math.pi


def finfo(dtype):
    return _torch.finfo(dtype)


# ---- devices and streams ----------------------------------------------------------------------------------------------
class Device:
    def __init__(self, kind: str):
        self.type = kind

    def __repr__(self):
        return f"Device({self.type})"


cpu, gpu = Device("cpu"), Device("gpu")


class Stream:
    def __init__(self, device: Device = gpu):
        self.device = device


_state = _NS(device="cuda" if _torch.cuda.is_available() else "cpu")


def default_device() -> Device:
    return gpu if _state.device == "cuda" else cpu


def set_default_device(device: Device) -> None:
    _state.device = "cuda" if device.type == "gpu" and _torch.cuda.is_available() else "cpu"


def default_stream(device: Device = gpu) -> Stream:
    return Stream(device)


def new_stream(device: Device = gpu) -> Stream:
    return Stream(device)


@contextlib.contextmanager
def stream(s):
    device = s.device if isinstance(s, Stream) else s
    previous = _state.device
    set_default_device(device)
    try:
        yield
    finally:
        _state.device = previous


def _sync():
    if _torch.cuda.is_available():
        _torch.cuda.synchronize()


def eval(*args):  # noqa: A001  (MLX's name)
    _sync()


def async_eval(*args):
    pass


def synchronize(*_):
    _sync()


def clear_cache():
    if _torch.cuda.is_available():
        _torch.cuda.empty_cache()


def device_info() -> dict:
    if _torch.cuda.is_available():
        p = _torch.cuda.get_device_properties(0)
        return {"device_name": p.name, "memory_size": p.total_memory, "architecture": getattr(p, "gcnArchName", "")}
    return {"device_name": "cpu", "memory_size": 0, "architecture": "cpu"}


def get_peak_memory() -> int:
    return _torch.cuda.max_memory_allocated() if _torch.cuda.is_available() else 0


def reset_peak_memory():
    if _torch.cuda.is_available():
        _torch.cuda.reset_peak_memory_stats()


# ---- torch.Tensor gains the MLX method names ---------------------------------------------------------------------------
def _astype(self, dtype, stream=None):
    return self.to(dtype)


class _At:
    def __init__(self, t):
        self._t = t

    def __getitem__(self, idx):
        t = self._t

        class _Op:
            def add(self, value):
                out = t.clone()
                out[idx] += value
                return out

            def multiply(self, value):
                out = t.clone()
                out[idx] *= value
                return out

        return _Op()


_torch_transpose = _torch.Tensor.transpose


def _transpose(self, *axes):
    if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
        axes = tuple(axes[0])
    if len(axes) == 0:
        return self.permute(*reversed(range(self.dim())))
    if len(axes) == self.dim() and len(axes) != 2:
        return self.permute(*axes)
    return _torch_transpose(self, *axes)


_torch_array = _torch.Tensor.__array__


def _array_protocol(self, dtype=None, *args, **kwargs):
    t = self.detach()
    if t.is_cuda:
        t = t.cpu()
    if t.dtype == _torch.bfloat16:
        t = t.float()
    return _torch_array(t, dtype) if dtype is not None else _torch_array(t)


_torch_size = _torch.Tensor.size


class _Size(int):
    """`a.size` is the element COUNT in MLX (an int property: benches/bench.py:297 `offset += context.size`) and a METHOD in
    torch (`t.size()`, `t.size(0)`).  One object serves both: an int that, when called, answers as torch's method does."""

    def __new__(cls, tensor):
        self = super().__new__(cls, tensor.numel())
        self._tensor = tensor
        return self

    def __call__(self, *args, **kwargs):
        return _torch_size(self._tensor, *args, **kwargs)


class _SizeProperty(property):
    __name__ = "size"  # torch.overrides enumerates Tensor attributes by __name__

    def __call__(self, tensor, *args, **kwargs):
        # tensor subclasses with __torch_function__ are handed `torch.Tensor.size` (looked up by name) to call
        with _torch._C.DisableTorchFunctionSubclass():
            return _torch_size(tensor, *args, **kwargs)


_torch.Tensor.size = _SizeProperty(_Size)
_torch.Tensor.astype = _astype
_torch.Tensor.at = property(lambda self: _At(self))
_torch.Tensor.transpose = _transpose
_torch.Tensor.__array__ = _array_protocol


# ---- construction ------------------------------------------------------------------------------------------------------
def _dev():
    return _state.device


def _make_array(value, dtype=None):
    if isinstance(value, _torch.Tensor):
        out = value.to(_dev())
        return out.to(dtype) if dtype is not None else out
    if isinstance(value, _np.ndarray):
        out = _torch.from_numpy(_np.ascontiguousarray(value)).to(_dev())
        return out.to(dtype) if dtype is not None else out
    if dtype is None:
        probe = _np.asarray(value)
        if probe.dtype.kind == "f":
            dtype = float32
        elif probe.dtype.kind in "iu":
            dtype = int32
        elif probe.dtype.kind == "b":
            dtype = bool_
    return _torch.tensor(value, dtype=dtype, device=_dev())


class _ArrayType(type):
    """`mx.array` is a TYPE in MLX: harness code writes `isinstance(mask, mx.array)` and `mx.array | str | None`."""

    def __instancecheck__(cls, obj):
        return isinstance(obj, _torch.Tensor)

    def __subclasscheck__(cls, sub):
        return issubclass(sub, _torch.Tensor)


class array(metaclass=_ArrayType):  # noqa: N801  (MLX's name)
    """`mx.array(value, dtype=None)` -> a torch.Tensor on the default device (arrays ARE tensors in this facade)."""

    def __new__(cls, value, dtype=None):
        return _make_array(value, dtype)


def zeros(shape, dtype=float32, stream=None):
    return _torch.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=dtype, device=_dev())


def ones(shape, dtype=float32, stream=None):
    return _torch.ones(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=dtype, device=_dev())


def full(shape, value, dtype=None, stream=None):
    return _torch.full(tuple(shape) if not isinstance(shape, int) else (shape,), value, dtype=dtype or float32, device=_dev())


def zeros_like(a):
    return _torch.zeros_like(a)


def ones_like(a):
    return _torch.ones_like(a)


def eye(n, dtype=float32):
    return _torch.eye(n, dtype=dtype, device=_dev())


def arange(*args, dtype=None, **kwargs):
    out = _torch.arange(*args, device=_dev())
    if dtype is not None:
        return out.to(dtype)
    return out.to(int32) if not out.is_floating_point() else out.to(float32)


def tril(a, k=0):
    return _torch.tril(a, diagonal=k)


def triu(a, k=0):
    return _torch.triu(a, diagonal=k)


# ---- elementwise / reductions --------------------------------------------------------------------------------------------
def _axis(kwargs):
    return kwargs.get("axis", None)


def _unary(fn):
    """MLX's element-wise functions also take Python scalars (`mx.rsqrt(self.head_dim)`, reference qwen3_week1.py:35) and
    return a 0-d float32 array for them."""

    def apply(a, stream=None):
        if not isinstance(a, _torch.Tensor):
            a = _torch.tensor(float(a), dtype=float32, device=_dev())
        return fn(a)

    apply.__name__ = fn.__name__
    return apply


exp, log, sin, cos, abs, sqrt, rsqrt, sigmoid, tanh, square, erf = (  # noqa: A001
    _unary(f) for f in (_torch.exp, _torch.log, _torch.sin, _torch.cos, _torch.abs, _torch.sqrt, _torch.rsqrt, _torch.sigmoid,
                        _torch.tanh, _torch.square, _torch.erf))
outer, maximum, minimum = _torch.outer, _torch.maximum, _torch.minimum


def matmul(a, b, stream=None):
    """MLX promotes mixed operands (a float16 activation on float32 weights gives float32: reference basics.py:18 on the Week-1
    tests' inputs); torch.matmul refuses them."""
    if a.dtype != b.dtype:
        common = _torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(common), b.to(common)
    return _torch.matmul(a, b)




def where(condition, x, y, stream=None):
    """MLX selects on any non-zero condition (`mx.where(mx.tril(mx.ones(...)), 0, -inf)`, reference attention.py:26) and
    promotes x / y like an arithmetic op."""
    if not isinstance(condition, _torch.Tensor):
        condition = _torch.tensor(condition, device=_dev())
    if condition.dtype != _torch.bool:
        condition = condition != 0
    if isinstance(x, _torch.Tensor) and isinstance(y, _torch.Tensor) and x.dtype != y.dtype:
        common = _torch.promote_types(x.dtype, y.dtype)
        x, y = x.to(common), y.to(common)
    return _torch.where(condition, x, y)


multiply, add, subtract, divide = _torch.mul, _torch.add, _torch.sub, _torch.div


def power(a, b):
    if not isinstance(a, _torch.Tensor):
        a = _torch.tensor(a, dtype=b.dtype if isinstance(b, _torch.Tensor) else float32, device=_dev())
    return _torch.pow(a, b)


def addmm(c, a, b, alpha=1.0, beta=1.0):
    return beta * c + alpha * _torch.matmul(a, b)


def _reduce(fn, a, axis=None, keepdims=False):
    if axis is None:
        return fn(a)
    return fn(a, dim=axis, keepdim=keepdims)


def sum(a, axis=None, keepdims=False):  # noqa: A001
    return _reduce(_torch.sum, a, axis, keepdims)


def mean(a, axis=None, keepdims=False):
    return _reduce(_torch.mean, a, axis, keepdims)


def max(a, axis=None, keepdims=False):  # noqa: A001
    return _torch.amax(a) if axis is None else _torch.amax(a, dim=axis, keepdim=keepdims)


def min(a, axis=None, keepdims=False):  # noqa: A001
    return _torch.amin(a) if axis is None else _torch.amin(a, dim=axis, keepdim=keepdims)


def argmax(a, axis=None, keepdims=False):
    return _torch.argmax(a) if axis is None else _torch.argmax(a, dim=axis, keepdim=keepdims)


def argsort(a, axis=-1):
    return _torch.argsort(a, dim=axis, stable=True)


def argpartition(a, kth, axis=-1):
    return _torch.argsort(a, dim=axis, stable=True)  # a full sort satisfies the partition contract


def cumsum(a, axis=None):
    return _torch.cumsum(a.reshape(-1) if axis is None else a, dim=0 if axis is None else axis)


def logsumexp(a, axis=None, keepdims=False):
    if axis is None:
        return _torch.logsumexp(a.reshape(-1), dim=0)
    return _torch.logsumexp(a, dim=axis, keepdim=keepdims)


def softmax(a, axis=-1, precise=False):
    if precise:
        return _torch.softmax(a.float(), dim=axis).to(a.dtype)
    return _torch.softmax(a, dim=axis)


def take_along_axis(a, indices, axis):
    return _torch.take_along_dim(a, indices.long(), dim=axis)


def expand_dims(a, axis):
    for ax in sorted(axis) if isinstance(axis, (tuple, list)) else (axis,):
        a = a.unsqueeze(ax)
    return a


def broadcast_to(a, shape):
    return _torch.broadcast_to(a, tuple(shape))


def repeat(a, repeats, axis=None):
    return _torch.repeat_interleave(a.reshape(-1) if axis is None else a, repeats, dim=0 if axis is None else axis)


def concatenate(arrays, axis=0):
    return _torch.cat(list(arrays), dim=axis)


concat = concatenate


def stack(arrays, axis=0):
    return _torch.stack(list(arrays), dim=axis)


def reshape(a, shape):
    return a.reshape(tuple(shape))


def transpose(a, axes=None):
    return a.permute(*reversed(range(a.dim()))) if axes is None else a.permute(*axes)


def swapaxes(a, a1, a2):
    return a.swapaxes(a1, a2)


def squeeze(a, axis=None):
    return a.squeeze() if axis is None else a.squeeze(axis)


def contiguous(a):
    return a.contiguous()


def array_equal(a, b, equal_nan=False):
    if a.shape != b.shape:
        return _torch.tensor(False)
    return _torch.tensor(bool(_torch.equal(a.to(b.device), b)))


def allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False):
    return _torch.tensor(bool(_torch.allclose(a.float(), b.float().to(a.device), rtol=rtol, atol=atol, equal_nan=equal_nan)))


def isnan(a):
    return _torch.isnan(a)


# ---- random -----------------------------------------------------------------------------------------------------------------
class _Random:
    def __init__(self):
        self._gen = {}

    def _g(self):
        dev = _dev()
        if dev not in self._gen:
            self._gen[dev] = _torch.Generator(device=dev)
            self._gen[dev].manual_seed(0)
        return self._gen[dev]

    def seed(self, value: int):
        for dev in ("cpu",) + (("cuda",) if _torch.cuda.is_available() else ()):
            self._gen[dev] = _torch.Generator(device=dev)
            self._gen[dev].manual_seed(int(value))

    def key(self, value: int):
        return int(value)

    def normal(self, shape=(), dtype=float32, loc=0.0, scale=1.0, key=None, stream=None):
        out = _torch.randn(tuple(shape), generator=self._g(), device=_dev(), dtype=_torch.float32) * scale + loc
        return out.to(dtype)

    def uniform(self, low=0.0, high=1.0, shape=(), dtype=float32, key=None, stream=None):
        out = _torch.rand(tuple(shape), generator=self._g(), device=_dev(), dtype=_torch.float32) * (high - low) + low
        return out.to(dtype)

    def randint(self, low, high, shape=(), dtype=int32, key=None, stream=None):
        return _torch.randint(int(low), int(high), tuple(shape), generator=self._g(), device=_dev()).to(dtype)

    def categorical(self, logits, axis=-1, shape=None, num_samples=None, key=None, stream=None):
        probs = _torch.softmax(logits.float(), dim=axis)
        flat = probs.reshape(-1, probs.shape[-1])
        return _torch.multinomial(flat, 1, generator=self._g()).reshape(probs.shape[:-1]).to(int32)


random = _Random()


# ---- quantisation oracles (parity unpinned against real MLX) -------------------------------------------
def quantize(w, group_size: int = 64, bits: int = 4, stream=None):
    from tiny_llm_hip.synthetic import quantize as _q

    return _q(w, group_size=group_size, bits=bits)


def _codes(packed, bits: int = 4):
    if bits != 4:
        raise ValueError("the facade restates 4-bit packing only")
    words = packed.to(_torch.int64) & 0xFFFFFFFF
    shifts = _torch.arange(0, 32, 4, device=packed.device, dtype=_torch.int64)
    return ((words.unsqueeze(-1) >> shifts) & 0xF).reshape(*packed.shape[:-1], packed.shape[-1] * 8)


def dequantize(w, scales, biases=None, group_size: int = 64, bits: int = 4, stream=None):
    q = _codes(w, bits).to(_torch.float32)
    s = scales.to(_torch.float32).repeat_interleave(group_size, dim=-1)
    out = q * s
    if biases is not None:
        out = out + biases.to(_torch.float32).repeat_interleave(group_size, dim=-1)
    return out.to(scales.dtype)


def quantized_matmul(x, w, scales, biases=None, transpose: bool = True, group_size: int = 64, bits: int = 4, stream=None):
    q = _codes(w, bits).to(_torch.float32)
    dense = q * scales.to(_torch.float32).repeat_interleave(group_size, dim=-1)
    if biases is not None:
        dense = dense + biases.to(_torch.float32).repeat_interleave(group_size, dim=-1)
    out = x.to(_torch.float32) @ (dense.transpose(-1, -2) if transpose else dense)
    return out.to(x.dtype)


# ---- mx.fast ------------------------------------------------------------------------------------------------------------------
def _fast_rms_norm(x, weight, eps: float, stream=None):
    xf = x.to(_torch.float32)
    out = xf * _torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    if weight is not None:
        out = out * weight.to(_torch.float32)
    return out.to(x.dtype)


def _fast_rope(x, dims: int, *, traditional: bool, base: float | None = 10000.0, scale: float = 1.0, offset=0, freqs=None,
               stream=None):
    """x [..., L, D] with the sequence on axis -2 (MLX layout [B, H, L, D]); offset int or [B] array."""
    L = x.shape[-2]
    half = dims // 2
    dev = x.device
    if freqs is None:
        inv = _torch.pow(_torch.tensor(float(base), device=dev), -_torch.arange(half, device=dev, dtype=_torch.float32) / half)
    else:
        inv = 1.0 / freqs.to(_torch.float32)
    if isinstance(offset, _torch.Tensor) and offset.dim() > 0:
        pos = offset.to(dev, _torch.float32).reshape(-1, *([1] * (x.dim() - 3)), 1) + _torch.arange(L, device=dev, dtype=_torch.float32)
        ang = pos.unsqueeze(-1) * scale * inv
    else:
        off = float(offset.item()) if isinstance(offset, _torch.Tensor) else float(offset)
        ang = ((off + _torch.arange(L, device=dev, dtype=_torch.float32)) * scale).unsqueeze(-1) * inv
    c, s = _torch.cos(ang), _torch.sin(ang)
    xf = x.to(_torch.float32)
    out = xf.clone()
    if traditional:
        re, im = xf[..., 0:dims:2], xf[..., 1:dims:2]
        out[..., 0:dims:2] = re * c - im * s
        out[..., 1:dims:2] = im * c + re * s
    else:
        re, im = xf[..., :half], xf[..., half:dims]
        out[..., :half] = re * c - im * s
        out[..., half:dims] = im * c + re * s
    return out.to(x.dtype)


def _fast_sdpa(q, k, v, *, scale: float, mask=None, stream=None):
    """[B, Hq, L, D] x [B, Hkv, S, D]: GQA by head grouping; mask additive array, boolean array or "causal" (lower-right)."""
    B, Hq, L, D = q.shape
    Hkv, S = k.shape[1], k.shape[2]
    rep = Hq // Hkv
    qf = q.to(_torch.float32).reshape(B, Hkv, rep, L, D)
    scores = _torch.matmul(qf, k.to(_torch.float32).unsqueeze(2).transpose(-1, -2)) * scale
    if isinstance(mask, str):
        if mask != "causal":
            raise ValueError(f"unsupported mask {mask!r}")
        keep = _torch.ones((L, S), dtype=_torch.bool, device=q.device).tril(diagonal=S - L)
        scores = scores.masked_fill(~keep, float("-inf"))
    elif mask is not None:
        m = mask.to(q.device)
        if m.dtype == _torch.bool:
            scores = scores.masked_fill(~m.reshape(*m.shape[:-2], 1, L, S) if m.dim() >= 4 else ~m, float("-inf"))
        else:
            mm = m.to(_torch.float32)
            if mm.dim() == 4:
                mm = mm.reshape(B, Hkv, rep, L, S) if mm.shape[1] == Hq else mm.unsqueeze(2)
            scores = scores + mm
    out = _torch.matmul(_torch.softmax(scores, dim=-1), v.to(_torch.float32).unsqueeze(2))
    return out.reshape(B, Hq, L, D).to(q.dtype)


fast = _NS(rms_norm=_fast_rms_norm, rope=_fast_rope, scaled_dot_product_attention=_fast_sdpa)


def gather_qmm(x, w, scales, biases=None, rhs_indices=None, transpose: bool = True, group_size: int = 64, bits: int = 4, **_):
    """x [..., M, N] times the expert matrices selected by rhs_indices (MoE helper; fp32 restatement)."""
    dense = dequantize(w, scales, biases, group_size, bits).to(_torch.float32)
    sel = dense[rhs_indices.long()]
    out = _torch.matmul(x.to(_torch.float32), sel.transpose(-1, -2) if transpose else sel)
    return out.to(x.dtype)
