"""Python faces of the fused Week-2 kernels (reference: src/tiny_llm_ref/week2_kernels.py)."""

import torch

from ._ext import tiny_llm_ext_hip
from .basics import softmax

# (1,) fp32 sentinel passed where the kernel takes a mask pointer it will not read (reference :7)
_NO_ATTENTION_MASK = None


def _no_mask(device) -> torch.Tensor:
    global _NO_ATTENTION_MASK
    if _NO_ATTENTION_MASK is None or _NO_ATTENTION_MASK.device != device:
        _NO_ATTENTION_MASK = torch.zeros((1,), dtype=torch.float32, device=device)
    return _NO_ATTENTION_MASK


class FastRMSNorm:
    """Single-rounding RMSNorm on the extension kernel (reference week2_kernels.py:10-19)."""

    def __init__(self, dim: int, weight: torch.Tensor, eps: float = 1e-5):
        self.dim = dim
        self.weight = weight
        self.eps = eps

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return tiny_llm_ext_hip.rms_norm(x.contiguous(), self.weight.to(x.dtype).contiguous(), self.eps)


class FastRoPE:
    """Kernel RoPE with one int32 offset per batch row (reference week2_kernels.py:22-53)."""

    def __init__(self, dims: int, seq_len: int, base: int = 10000, traditional: bool = False):
        self.dims = dims
        self.seq_len = seq_len
        self.base = base
        self.traditional = traditional

    def __call__(self, x: torch.Tensor, offset: int | list[int] | torch.Tensor = 0) -> torch.Tensor:
        batch = x.shape[0]
        if isinstance(offset, int):
            offsets = torch.full((batch,), offset, dtype=torch.int32, device=x.device)
        elif isinstance(offset, list):
            if len(offset) != batch:
                raise ValueError("FastRoPE needs one offset per batch row")
            offsets = torch.tensor(offset, dtype=torch.int32, device=x.device)
        elif offset.dim() == 0:
            offsets = offset.to(device=x.device, dtype=torch.int32).expand(batch)
        elif tuple(offset.shape) != (batch,):
            raise ValueError("FastRoPE needs one offset per batch row")
        else:
            offsets = offset.to(device=x.device, dtype=torch.int32)
        return tiny_llm_ext_hip.rope(x.contiguous(), offsets.contiguous(), self.dims, self.base, self.traditional)


def swiglu(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    return tiny_llm_ext_hip.swiglu(gate.contiguous(), up.contiguous())


def scaled_dot_product_attention(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    scale: float,
    mask: torch.Tensor | str | None = None,
) -> torch.Tensor:
    """Readable grouped-query attention that stays in the input dtype (Week-3 dense fallback,
    reference week2_kernels.py:60-95)."""
    shape = query.shape
    lead = query.shape[:-3]
    heads, q_len, dim = query.shape[-3:]
    kv_heads, ctx, _ = key.shape[-3:]
    if key.shape != value.shape or heads % kv_heads != 0:
        raise ValueError("incompatible grouped-query attention shapes")
    rep = heads // kv_heads
    q = query.reshape(*lead, kv_heads, rep, q_len, dim)
    k = key.reshape(*lead, kv_heads, 1, ctx, dim)
    v = value.reshape(*lead, kv_heads, 1, ctx, dim)
    scores = torch.matmul(q, k.transpose(-2, -1)) * torch.tensor(scale, dtype=query.dtype, device=query.device)
    if isinstance(mask, str):
        if mask != "causal":
            raise ValueError(f"unsupported attention mask: {mask}")
        keep = torch.tril(torch.ones((q_len, ctx), dtype=torch.bool, device=query.device), diagonal=ctx - q_len)
        scores = scores + torch.where(keep, 0.0, float("-inf")).to(scores.dtype)
    elif mask is not None:
        full = torch.broadcast_to(mask, (*lead, heads, q_len, ctx))
        scores = scores + full.reshape(*lead, kv_heads, rep, q_len, ctx)
    return torch.matmul(softmax(scores, axis=-1), v).reshape(shape)


def decode_attention_custom(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    scale: float,
    mask: torch.Tensor | str | None = None,
) -> torch.Tensor:
    """Dense-KV decode attention kernel behind the [B, H, L, D] model layout (reference week2_kernels.py:98-147)."""
    batch, heads, q_len, dim = query.shape
    k_batch, kv_heads, ctx, k_dim = key.shape
    if batch != k_batch or key.shape != value.shape:
        raise ValueError("query, key, and value batch dimensions must match")
    if dim != k_dim or heads % kv_heads != 0:
        raise ValueError("incompatible grouped-query attention shapes")
    if isinstance(mask, str) and mask != "causal":
        raise ValueError(f"unsupported attention mask: {mask}")
    q = query.reshape(batch * heads, q_len, dim).contiguous()
    k = key.reshape(batch * kv_heads, ctx, dim).contiguous()
    v = value.reshape(batch * kv_heads, ctx, dim).contiguous()
    causal = isinstance(mask, str)
    explicit = isinstance(mask, torch.Tensor)
    if explicit:
        dense = torch.broadcast_to(mask, (batch, heads, q_len, ctx)).to(torch.float32)
        mask_arg = dense.reshape(batch * heads, q_len, ctx).contiguous()
    else:
        mask_arg = _no_mask(query.device)
    out = tiny_llm_ext_hip.decode_attention(q, k, v, mask_arg, scale, causal, explicit, heads, kv_heads)
    return out.reshape(batch, heads, q_len, dim)
