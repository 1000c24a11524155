"""Readable building blocks on torch tensors (reference: src/tiny_llm_ref/basics.py)."""

import torch


def softmax(x: torch.Tensor, axis: int) -> torch.Tensor:
    """The library softmax, as in the reference (basics.py:5-7: ``mx.softmax(x, axis=axis)``) -- 16-bit inputs are reduced in
    float32 and rounded once -- except that a row that is entirely -inf (an idle row of a batch whose mask hides every key)
    comes out as zeros instead of NaN."""
    out = torch.softmax(x, dim=axis)
    dead = torch.isneginf(x).all(dim=axis, keepdim=True)
    return torch.where(dead, torch.zeros_like(out), out)


def linear(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """``x @ w.T (+ bias)`` with w stored [out, in] (reference basics.py:10-18).  Mixed float operands are promoted
    as ``mx.matmul`` promotes them (f16 activations on f32 weights -> f32: tests_refsol/test_week_1_day_3.py:171-199)."""
    if x.dtype != w.dtype:
        common = torch.promote_types(x.dtype, w.dtype)
        x, w = x.to(common), w.to(common)
    y = torch.matmul(x, w.transpose(-1, -2))
    return y if bias is None else y + bias


def silu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(x), evaluated through exp(-|x|) so large negative inputs do not overflow
    (reference basics.py:21-26)."""
    z = torch.exp(-torch.abs(x))
    return x * torch.where(x < 0, z / (1 + z), 1 / (1 + z))
