"""Attention operators (reference: src/tiny_llm_ref/attention.py)."""

import torch

from ._ext import tiny_llm_ext_hip
from .basics import linear, softmax


def scaled_dot_product_attention_simple(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    scale: float | None = None,
    mask: torch.Tensor | None = None,
) -> torch.Tensor:
    """softmax(q k^T * scale + mask) v for equally-shaped q/k/v (reference attention.py:6-21)."""
    factor = query.shape[-1] ** -0.5 if scale is None else scale
    scores = torch.matmul(query, key.transpose(-2, -1)) * factor
    if mask is not None:
        scores = scores + mask
    return torch.matmul(softmax(scores, axis=-1), value)


def causal_mask(L: int, S: int, dtype: torch.dtype, device=None) -> torch.Tensor:
    """[L, S] additive mask whose diagonal is aligned to the END of the context (reference attention.py:24-27)."""
    keep = torch.tril(torch.ones((L, S), dtype=torch.bool, device=device), diagonal=S - L)
    zero = torch.zeros((), dtype=dtype, device=device)
    return torch.where(keep, zero, torch.full((), float("-inf"), dtype=dtype, device=device))


def scaled_dot_product_attention_grouped(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    scale: float | None = None,
    mask: torch.Tensor | str | None = None,
) -> torch.Tensor:
    """Readable grouped-query attention: query heads are folded to [.., Hkv, rep, L, D] and broadcast against
    [.., Hkv, 1, S, D] keys/values; the full L x S score matrix is materialised (reference attention.py:30-66).
    ``mask`` is None, "causal", or an additive tensor broadcastable to [.., Hq, L, S]."""
    shape = query.shape
    heads, q_len, dim = query.shape[-3:]
    kv_heads, ctx, _ = key.shape[-3:]
    lead = query.shape[:-3]
    assert heads % kv_heads == 0
    rep = heads // kv_heads
    factor = torch.tensor(dim ** -0.5 if scale is None else float(scale), dtype=query.dtype, device=query.device)
    q = query.reshape(*lead, kv_heads, rep, q_len, dim)
    k = key.reshape(*lead, kv_heads, 1, ctx, dim)
    v = value.reshape(*lead, kv_heads, 1, ctx, dim)
    scores = torch.matmul(q, k.transpose(-2, -1)) * factor
    if isinstance(mask, str):
        if mask != "causal":
            raise ValueError(f"unsupported attention mask: {mask}")
        scores = scores + causal_mask(q_len, ctx, scores.dtype, scores.device)
    elif mask is not None:
        full = torch.broadcast_to(mask, (*lead, heads, q_len, ctx))
        scores = scores + full.reshape(*lead, kv_heads, rep, q_len, ctx)
    probs = softmax(scores, axis=-1)
    # an fp32 mask on 16-bit inputs promotes the scores (as MLX's type promotion does); the values follow them
    return torch.matmul(probs, v.to(probs.dtype)).reshape(shape)


def _validate_page_metadata(context_values, block_rows, *, page_size, max_pages, num_physical_pages, L) -> None:
    """Host-side metadata checks of the reference wrapper (attention.py:133-158); messages are part of the API."""
    seen: set[int] = set()
    for b, (ctx, row) in enumerate(zip(context_values, block_rows)):
        if ctx < 0:
            raise ValueError(f"context_lens[{b}] must be nonnegative")
        live = (ctx + page_size - 1) // page_size
        if live > max_pages:
            raise ValueError(f"context_lens[{b}] is not covered by block_table")
        for lp, page_id in enumerate(row):
            if lp < live:
                if page_id < 0 or page_id >= num_physical_pages:
                    raise ValueError(
                        f"Live page id {page_id} at [{b}, {lp}] is outside physical page storage"
                    )
                if page_id in seen:
                    raise ValueError(f"Live page id {page_id} is aliased")
                seen.add(page_id)
            elif page_id != -1:
                raise ValueError(f"Unused block_table entry [{b}, {lp}] must use the -1 sentinel")
        if 0 < ctx < L:
            raise ValueError(f"context_lens[{b}] must be zero or at least query length {L}")


def paged_attention(
    query: torch.Tensor,
    key_pages: torch.Tensor,
    value_pages: torch.Tensor,
    block_table: torch.Tensor,
    context_lens: torch.Tensor,
    page_size: int,
    scale: float | None = None,
    mask: torch.Tensor | str | None = None,
    *,
    host_block_rows: list[list[int]] | None = None,
    host_context_lens: list[int] | None = None,
) -> torch.Tensor:
    """Attention straight from paged K/V storage, model layout [B, Hq, L, D] (reference attention.py:69-178).

    The reference pulls ``block_table``/``context_lens`` back to the host on every call to validate them
    (two device->host syncs per layer per step).  Callers that built the metadata on the host (the KV caches
    do) pass the same integers through ``host_block_rows`` / ``host_context_lens`` so the checks run with
    no synchronisation; without them the tensors are read back exactly like the reference."""
    if isinstance(mask, torch.Tensor):
        raise NotImplementedError("Paged attention only supports mask=None or causal")
    if mask is not None and mask != "causal":
        raise NotImplementedError
    if query.dim() != 4:
        raise ValueError("query must be 4D [B, H_q, L, D]")
    if key_pages.dim() != 4 or value_pages.dim() != 4:
        raise ValueError("page tensors must be 4D [P, H_kv, page_size, D]")
    if key_pages.shape != value_pages.shape:
        raise ValueError("key pages and value pages must have the same shape")
    if block_table.dim() != 2 or context_lens.dim() != 1:
        raise ValueError("block_table must be 2D and context_lens must be 1D")
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise ValueError("block_table and context_lens must be int32")
    if not isinstance(page_size, int) or page_size <= 0:
        raise ValueError("page_size must be a positive integer")

    B, heads, L, D = query.shape
    physical, kv_heads, stored_page, stored_dim = key_pages.shape
    if min(B, heads, L, D, kv_heads, stored_page, stored_dim) <= 0:
        raise ValueError("paged attention dimensions must be positive")
    if physical <= 0:
        raise ValueError("paged attention requires nonempty physical page storage")
    if heads % kv_heads != 0:
        raise ValueError("query heads must be divisible by K/V heads")
    if stored_dim != D:
        raise ValueError("query and page tensors must have the same head dimension")
    if stored_page != page_size:
        raise ValueError(f"page_size={page_size} does not match page storage {stored_page}")
    if block_table.shape[0] != B or context_lens.shape[0] != B:
        raise ValueError("query, block_table, and context_lens batch sizes must match")
    max_pages = block_table.shape[1]
    if max_pages <= 0:
        raise ValueError("block_table must provide at least one page slot")
    if query.dtype != key_pages.dtype or query.dtype != value_pages.dtype:
        raise ValueError("query, key pages, and value pages must have the same dtype")
    if query.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("paged attention supports float32 or bfloat16 inputs")

    context_values = host_context_lens if host_context_lens is not None else context_lens.tolist()
    block_rows = host_block_rows if host_block_rows is not None else block_table.tolist()
    _validate_page_metadata(
        context_values, block_rows, page_size=page_size, max_pages=max_pages, num_physical_pages=physical, L=L
    )

    factor = D ** -0.5 if scale is None else float(scale)
    out = tiny_llm_ext_hip.paged_attention(
        query.reshape(B * heads, L, D).contiguous(),
        key_pages.contiguous(),
        value_pages.contiguous(),
        block_table.contiguous(),
        context_lens.contiguous(),
        factor,
        is_causal=(mask == "causal"),
        num_kv_heads=kv_heads,
        num_heads=heads,
        max_context_hint=max(context_values, default=0),
    )
    return out.reshape(B, heads, L, D).contiguous()


class SimpleMultiHeadAttention:
    """Week-1 multi-head attention over dense fp weights (reference attention.py:181-237)."""

    def __init__(self, hidden_size: int, num_heads: int, wq, wk, wv, wo):
        assert hidden_size % num_heads == 0
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5
        for w in (wq, wk, wv):
            assert tuple(w.shape) == (num_heads * self.head_dim, hidden_size)
        assert tuple(wo.shape) == (hidden_size, num_heads * self.head_dim)
        self.wq, self.wk, self.wv, self.wo = wq, wk, wv, wo

    def _split(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        n, length, _ = x.shape
        return linear(x, w).reshape(n, length, self.num_heads, self.head_dim).transpose(1, 2)

    def __call__(self, query, key, value, mask: torch.Tensor | None = None) -> torch.Tensor:
        assert query.shape == key.shape == value.shape
        n, length, _ = query.shape
        mixed = scaled_dot_product_attention_simple(
            self._split(query, self.wq), self._split(key, self.wk), self._split(value, self.wv),
            scale=self.scale, mask=mask,
        )
        return linear(mixed.transpose(1, 2).reshape(n, length, self.hidden_size), self.wo)
