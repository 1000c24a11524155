"""Readable RMSNorm (reference: src/tiny_llm_ref/layer_norm.py:4-15)."""

import torch


class RMSNorm:
    """Two-rounding order of the readable reference: normalise in fp32, cast back, then scale by the weight
    in the activation dtype.  The fused kernel (``FastRMSNorm``) rounds once instead."""

    def __init__(self, dim: int, weight: torch.Tensor, eps: float = 1e-5):
        self.dim = dim
        self.weight = weight
        self.eps = eps

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        wide = x.to(torch.float32)
        wide = wide * torch.rsqrt(wide.square().mean(dim=-1, keepdim=True) + self.eps)
        return wide.to(x.dtype) * self.weight.to(x.dtype)
