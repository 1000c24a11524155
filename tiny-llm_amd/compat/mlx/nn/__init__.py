"""`mlx.nn` names the reference's tests and benches import (facade over torch; see ../../README.md).

The reference's Week-1 and MoE tests use MLX's own layers as their ORACLES (`nn.MultiHeadAttention`,
tests_refsol/test_week_1_day_1.py:152; `nn.Linear(...).to_quantized`, `nn.quantize`, test_week_3_day_6.py:58-118;
`nn.init.he_uniform`, benches/test_attention.py:10).  They are restated here in fp32 torch from MLX's documented
semantics — PARITY UNPINNED against real MLX, which cannot be installed in this image.
Arrays are plain torch tensors; a layer is a plain Python object whose parameters are attributes, so the tests'
`layer.weight = ...` assignments work as they do on an `mlx.nn.Module`.
"""

from __future__ import annotations

import math as _math
from types import SimpleNamespace as _NS

import torch as _torch

from .. import core as _mx


def silu(x):
    return x * _torch.sigmoid(x)


def gelu(x):
    return _torch.nn.functional.gelu(x)


def relu(x):
    return _torch.relu(x)


def _common(*tensors):
    """MLX promotes mixed float operands (f16 x f32 -> f32); torch.matmul refuses them."""
    dtype = tensors[0].dtype
    for t in tensors[1:]:
        dtype = _torch.promote_types(dtype, t.dtype)
    return [t.to(dtype) for t in tensors]


class Module:
    """The slice of `mlx.nn.Module` the tests touch: attribute parameters, child discovery, `__getitem__`."""

    def __getitem__(self, name):
        return getattr(self, name)

    def __contains__(self, name):
        return name in self.__dict__

    def get(self, name, default=None):
        return self.__dict__.get(name, default)

    def children(self):
        return {k: v for k, v in self.__dict__.items() if isinstance(v, Module)}

    def named_modules(self, prefix=""):
        yield prefix, self
        for name, child in self.children().items():
            yield from child.named_modules(f"{prefix}.{name}" if prefix else name)

    def parameters(self):
        out = {}
        for k, v in self.__dict__.items():
            if isinstance(v, _torch.Tensor):
                out[k] = v
            elif isinstance(v, Module):
                out[k] = v.parameters()
        return out


class Linear(Module):
    """y = x W^T + b, weight [out, in] drawn from U(-1/sqrt(in), 1/sqrt(in))."""

    def __init__(self, input_dims: int, output_dims: int, bias: bool = True):
        scale = _math.sqrt(1.0 / input_dims)
        self.weight = _mx.random.uniform(-scale, scale, shape=(output_dims, input_dims))
        if bias:
            self.bias = _mx.random.uniform(-scale, scale, shape=(output_dims,))

    def __call__(self, x):
        x, w = _common(x, self.weight)
        out = x @ w.transpose(-1, -2)
        if "bias" in self:
            out = out + self.bias.to(out.dtype)
        return out

    def to_quantized(self, group_size: int = 64, bits: int = 4, mode: str = "affine"):
        return QuantizedLinear.from_linear(self, group_size, bits)


class QuantizedLinear(Module):
    def __init__(self, weight, scales, biases, group_size: int, bits: int, bias=None):
        self.weight, self.scales, self.biases = weight, scales, biases
        self.group_size, self.bits = group_size, bits
        if bias is not None:
            self.bias = bias

    @classmethod
    def from_linear(cls, linear: Linear, group_size: int = 64, bits: int = 4):
        w, s, b = _mx.quantize(linear.weight, group_size=group_size, bits=bits)
        return cls(w, s, b, group_size, bits, bias=linear.get("bias"))

    def __call__(self, x):
        out = _mx.quantized_matmul(x, self.weight, self.scales, self.biases, transpose=True,
                                   group_size=self.group_size, bits=self.bits)
        if "bias" in self:
            out = out + self.bias.to(out.dtype)
        return out


class QuantizedEmbedding(Module):
    """Rows of a W4 table, dequantised on lookup; `as_linear` is the tied output projection (mlx.nn.QuantizedEmbedding)."""

    def __init__(self, weight, scales, biases, group_size: int, bits: int):
        self.weight, self.scales, self.biases = weight, scales, biases
        self.group_size, self.bits = group_size, bits
        self.num_embeddings, self.dims = weight.shape[0], weight.shape[1] * 32 // bits

    def __call__(self, ids):
        flat = ids.reshape(-1).long()
        rows = _mx.dequantize(self.weight[flat], self.scales[flat], self.biases[flat], self.group_size, self.bits)
        return rows.reshape(*ids.shape, -1)

    def as_linear(self, x):
        return _mx.quantized_matmul(x, self.weight, self.scales, self.biases, transpose=True,
                                    group_size=self.group_size, bits=self.bits)


class RMSNorm(Module):
    def __init__(self, dims: int, eps: float = 1e-5):
        self.weight = _mx.ones((dims,))
        self.eps = eps

    def __call__(self, x):
        return _mx.fast.rms_norm(x, self.weight, self.eps)


class MultiHeadAttention(Module):
    """Four bias-free projections around `mx.fast.scaled_dot_product_attention` with scale 1/sqrt(head_dim);
    inputs [B, L, dims], mask additive and broadcast over batch and heads."""

    def __init__(self, dims: int, num_heads: int, query_input_dims=None, key_input_dims=None, value_input_dims=None,
                 value_dims=None, value_output_dims=None, bias: bool = False):
        if dims % num_heads != 0:
            raise ValueError(f"The input feature dimensions should be divisible by the number of heads ({dims} % {num_heads}) != 0")
        value_dims = value_dims or dims
        self.num_heads = num_heads
        self.query_proj = Linear(query_input_dims or dims, dims, bias=bias)
        self.key_proj = Linear(key_input_dims or dims, dims, bias=bias)
        self.value_proj = Linear(value_input_dims or key_input_dims or dims, value_dims, bias=bias)
        self.out_proj = Linear(value_dims, value_output_dims or dims, bias=bias)

    def __call__(self, queries, keys, values, mask=None):
        q, k, v = self.query_proj(queries), self.key_proj(keys), self.value_proj(values)
        H = self.num_heads
        B, L, _ = q.shape
        S = k.shape[1]
        q = q.reshape(B, L, H, -1).permute(0, 2, 1, 3)
        k = k.reshape(B, S, H, -1).permute(0, 2, 1, 3)
        v = v.reshape(B, S, H, -1).permute(0, 2, 1, 3)
        q, k, v = _common(q, k, v)
        out = _mx.fast.scaled_dot_product_attention(q, k, v, scale=_math.sqrt(1.0 / q.shape[-1]), mask=mask)
        return self.out_proj(out.permute(0, 2, 1, 3).reshape(B, L, -1))

    @staticmethod
    def create_additive_causal_mask(N: int, dtype=_mx.float32):
        idx = _torch.arange(N)
        return (idx[:, None] < idx[None]).to(dtype) * _torch.finfo(dtype).min


def quantize(model: Module, group_size: int = 64, bits: int = 4, class_predicate=None, mode: str = "affine"):
    """Replace, in place, every child layer that has `to_quantized` (and passes `class_predicate(path, layer)`)."""
    class_predicate = class_predicate or (lambda _path, m: hasattr(m, "to_quantized"))

    def walk(parent: Module, prefix: str):
        for name, child in list(parent.children().items()):
            path = f"{prefix}.{name}" if prefix else name
            if hasattr(child, "to_quantized") and class_predicate(path, child):
                setattr(parent, name, child.to_quantized(group_size=group_size, bits=bits))
            else:
                walk(child, path)

    walk(model, "")
    return model


def _he_uniform(dtype=_mx.float32):
    """`nn.init.he_uniform(dtype)(array)`: U(-limit, limit) with limit = gain * sqrt(3 / fan_in), array shaped like the input."""

    def initializer(a, mode: str = "fan_in", gain: float = 1.0):
        fan_out, fan_in = (a.shape[0], a.shape[-1]) if a.dim() > 1 else (a.shape[0], a.shape[0])
        if a.dim() > 2:
            receptive = _math.prod(a.shape[1:-1])
            fan_in, fan_out = fan_in * receptive, fan_out * receptive
        limit = gain * _math.sqrt(3.0 / (fan_in if mode == "fan_in" else fan_out))
        return _mx.random.uniform(-limit, limit, shape=tuple(a.shape), dtype=dtype)

    return initializer


init = _NS(he_uniform=_he_uniform)
