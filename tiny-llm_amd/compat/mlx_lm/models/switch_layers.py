"""`mlx_lm.models.switch_layers`: the grouped-expert layers the reference's optional MoE chapter checks against
(tests_refsol/test_week_3_day_6.py:10,36-50).  Restated over the facade's `mx.gather_qmm` (fp32 torch); PARITY UNPINNED
against real mlx-lm.  Weights are [experts, out, in]; `indices` selects an expert per batch position of `x[..., M, in]`.
"""

from __future__ import annotations

import math

import torch

import mlx.core as mx
import mlx.nn as nn


class QuantizedSwitchLinear(nn.Module):
    def __init__(self, weight, scales, biases, group_size: int, bits: int, bias=None):
        self.weight, self.scales, self.biases = weight, scales, biases
        self.group_size, self.bits = group_size, bits
        if bias is not None:
            self.bias = bias

    def __call__(self, x, indices, sorted_indices: bool = False):
        out = mx.gather_qmm(x, self.weight, self.scales, self.biases, rhs_indices=indices, transpose=True,
                            group_size=self.group_size, bits=self.bits)
        if "bias" in self:
            out = out + self.bias[indices.long()].unsqueeze(-2).to(out.dtype)
        return out


class SwitchLinear(nn.Module):
    def __init__(self, input_dims: int, output_dims: int, num_experts: int, bias: bool = True):
        scale = math.sqrt(1.0 / input_dims)
        self.weight = mx.random.uniform(-scale, scale, shape=(num_experts, output_dims, input_dims))
        if bias:
            self.bias = mx.zeros((num_experts, output_dims))

    def __call__(self, x, indices, sorted_indices: bool = False):
        w = self.weight[indices.long()]
        dtype = torch.promote_types(x.dtype, w.dtype)
        out = torch.matmul(x.to(dtype), w.to(dtype).transpose(-1, -2))
        if "bias" in self:
            out = out + self.bias[indices.long()].unsqueeze(-2).to(out.dtype)
        return out

    def to_quantized(self, group_size: int = 64, bits: int = 4, mode: str = "affine"):
        w, s, b = mx.quantize(self.weight, group_size=group_size, bits=bits)
        return QuantizedSwitchLinear(w, s, b, group_size, bits, bias=self.get("bias"))


class SwitchGLU(nn.Module):
    def __init__(self, input_dims: int, hidden_dims: int, num_experts: int, activation=None, bias: bool = False):
        self.gate_proj = SwitchLinear(input_dims, hidden_dims, num_experts, bias=bias)
        self.up_proj = SwitchLinear(input_dims, hidden_dims, num_experts, bias=bias)
        self.down_proj = SwitchLinear(hidden_dims, input_dims, num_experts, bias=bias)
        self.activation = activation or (lambda up, gate: nn.silu(gate) * up)

    def __call__(self, x, indices):
        x = x.unsqueeze(-2).unsqueeze(-3)  # [..., 1, 1, D]: one row per (token, selected expert) after broadcasting
        up = self.up_proj(x, indices)
        gate = self.gate_proj(x, indices)
        return self.down_proj(self.activation(up, gate), indices).squeeze(-2)
