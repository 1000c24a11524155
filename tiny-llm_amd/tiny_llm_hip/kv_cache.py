"""Dense KV caches and the batching cache (reference: src/tiny_llm_ref/kv_cache.py)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import TYPE_CHECKING, Optional

import torch

from .attention import causal_mask

if TYPE_CHECKING:  # pragma: no cover
    from .paged_kv_cache import PagedKvMetadata


def materialize_tensors(*tensors: torch.Tensor) -> None:
    """Counterpart of ``mx.eval(...)`` on cache storage.  PyTorch launches eagerly, so there is no lazy graph
    to cut; the hook exists so schedulers keep the reference's call structure (and tests can observe it).
    When the `mlx.core` import facade (tiny-llm_amd/compat) is loaded, the call goes through ITS ``eval`` -- harness code
    written against the reference instruments ``mx.eval`` (tests_refsol/test_week_3_day_3.py:289-306)."""
    import sys

    facade = sys.modules.get("mlx.core")
    if facade is not None and hasattr(facade, "eval"):
        facade.eval(*tensors)
    return None


class TinyKvCache(ABC):
    """Per-layer KV cache protocol (reference kv_cache.py:11-70)."""

    @abstractmethod
    def update_and_fetch(
        self,
        key: torch.Tensor,
        value: torch.Tensor,
        mask_length: int | None = None,
        mask: torch.Tensor | str | None = None,
    ) -> tuple[torch.Tensor, torch.Tensor, int, Optional[torch.Tensor]]:
        """Append ``key``/``value`` [B, H, S, D]; return (all keys, all values, sequence length, mask)."""

    def release(self):
        """Give back whatever the cache owns (pages for paged caches; nothing for dense ones)."""
        return None

    def materialize(self):
        """Force evaluation of the owned storage without changing its layout."""
        return None

    def update_and_fetch_paged(self, key, value, mask_length=None, mask=None) -> "PagedKvMetadata":
        raise NotImplementedError("This KV cache does not support paged attention")

    def rewind(self, n: int):
        """Drop the newest ``n`` tokens (speculative decoding)."""
        raise NotImplementedError("This KV cache does not support rewind")


class TinyKvFullCache(TinyKvCache):
    """Growing dense cache: every append concatenates (O(S) copy per step), as the reference does
    (kv_cache.py:246-287).  ``growth_copy_bytes`` counts the bytes re-copied by those concatenations."""

    def __init__(self):
        self.key_values: tuple[torch.Tensor, torch.Tensor] | None = None
        self.offset = 0
        self.growth_copy_bytes = 0

    def update_and_fetch(self, key, value, mask_length=None, mask=None):
        if self.key_values is None:
            assert self.offset == 0
            self.key_values = (key, value)
            self.offset = key.shape[2]
            return key, value, self.offset, mask
        B, H, S, D = key.shape
        assert key.shape == value.shape
        old_k, old_v = self.key_values
        assert tuple(old_k.shape) == (B, H, self.offset, D)
        assert tuple(old_v.shape) == (B, H, self.offset, D)
        self.growth_copy_bytes += old_k.numel() * old_k.element_size() + old_v.numel() * old_v.element_size()
        self.key_values = (torch.cat([old_k, key], dim=2), torch.cat([old_v, value], dim=2))
        self.offset += S
        return self.key_values[0], self.key_values[1], self.offset, mask

    def materialize(self):
        if self.key_values is not None:
            materialize_tensors(*self.key_values)

    def rewind(self, n: int):
        self.offset -= n
        self.key_values = (self.key_values[0][:, :, : self.offset], self.key_values[1][:, :, : self.offset])


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


class BatchingKvCache(TinyKvCache):
    """One slot per concurrently decoding request (reference kv_cache.py:73-243).

    ``update_and_fetch`` rebuilds a right-aligned dense batch (Week-3 day-1 behaviour);
    ``update_and_fetch_paged`` appends each active row to its paged request cache and returns block-table
    metadata, all-or-nothing: any failure restores the pool and every request cache."""

    def __init__(self, max_active_requests: int, max_seq_len: int | None = None):
        self.max_active_requests = max_active_requests
        self.max_seq_len = max_seq_len
        self.kv_caches: list[TinyKvCache | None] = [None] * max_active_requests
        self.HD = None
        self.last_batch_bytes = 0
        self.staging_copy_bytes = 0

    def update_and_fetch(self, keys, values, mask_length=None, mask=None):
        B, H, S, D = keys.shape
        assert keys.shape == values.shape
        if self.max_seq_len is not None:
            assert S <= self.max_seq_len
        if self.HD is None:
            self.HD = (H, D)
        else:
            assert self.HD == (H, D), f"expect {self.HD} but got {H, D}"
        assert B == self.max_active_requests
        rows: list[tuple | None] = []
        for b, cache in enumerate(self.kv_caches):
            if cache is None:
                rows.append(None)
                continue
            k, v, length, row_mask = cache.update_and_fetch(keys[b : b + 1], values[b : b + 1])
            rows.append((k[0], v[0], length, row_mask))
        seq_len = max((r[2] for r in rows if r is not None), default=0)
        dense_k = torch.zeros((B, H, seq_len, D), dtype=keys.dtype, device=keys.device)
        dense_v = torch.zeros((B, H, seq_len, D), dtype=values.dtype, device=values.device)
        masks = torch.full((B, mask_length, seq_len), float("-inf"), dtype=keys.dtype, device=keys.device)
        for b, row in enumerate(rows):
            if row is None:
                continue
            k, v, length, row_mask = row
            self.staging_copy_bytes += _nbytes(k) + _nbytes(v)
            dense_k[b, :, seq_len - length :, :] = k
            dense_v[b, :, seq_len - length :, :] = v
            if row_mask is None or (isinstance(row_mask, str) and row_mask == "causal"):
                masks[b, :, seq_len - length :] = causal_mask(mask_length, length, keys.dtype, keys.device)
            elif isinstance(row_mask, torch.Tensor):
                masks[b, :, seq_len - length :] = row_mask
            else:
                raise NotImplementedError
        self.last_batch_bytes = _nbytes(dense_k) + _nbytes(dense_v)
        return dense_k, dense_v, None, masks.reshape(B, 1, mask_length, seq_len)

    def update_and_fetch_paged(self, keys, values, mask_length=None, mask=None):
        from .paged_kv_cache import PagedKvMetadata, TinyKvPagedCache

        if keys.dim() != 4 or values.dim() != 4:
            raise ValueError("Batched K/V chunks must be 4D [B, H, S, D]")
        if keys.shape != values.shape:
            raise ValueError("Batched K/V chunks must have the same shape")
        B, H, S, D = keys.shape
        if B != self.max_active_requests:
            raise ValueError(f"Expected batch size {self.max_active_requests}, got {B}")
        if self.HD is not None and self.HD != (H, D):
            raise ValueError(f"expect {self.HD} but got {H, D}")

        # Phase 1: validate every active row before anything is mutated.
        pool = None
        active: list[tuple[int, TinyKvPagedCache]] = []
        for b, cache in enumerate(self.kv_caches):
            if cache is None:
                continue
            if not isinstance(cache, TinyKvPagedCache):
                raise ValueError("BatchingKvCache contains a non-paged request cache")
            if pool is None:
                pool = cache.pool
            elif cache.pool is not pool:
                raise ValueError("Paged batch caches must share one page pool")
            if self.max_seq_len is not None and cache.offset + S > self.max_seq_len:
                raise ValueError("Paged batch append exceeds max_seq_len")
            cache.validate_append(keys[b : b + 1], values[b : b + 1])
            active.append((b, cache))
        if pool is None:
            raise ValueError("Cannot build paged metadata without active requests")

        # Phase 2: append row by row inside one transaction.
        pool_before = pool._snapshot_state()
        caches_before = [(cache, cache._snapshot_state()) for _, cache in active]
        hd_before = self.HD
        context = [0] * B
        widest = 0
        try:
            for b, cache in active:
                cache.update_and_fetch_paged(keys[b : b + 1], values[b : b + 1], mask_length=mask_length, mask=mask)
                context[b] = cache.offset
                widest = max(widest, cache.num_pages)
            self.HD = (H, D)
        except Exception:
            pool._restore_state(pool_before)
            for cache, state in caches_before:
                cache._restore_state(state)
            self.HD = hd_before
            raise

        self.last_batch_bytes = 0
        table = [
            [-1] * widest if cache is None else cache.page_ids + [-1] * (widest - cache.num_pages)
            for cache in self.kv_caches
        ]
        device = pool.key_pages.device
        return PagedKvMetadata(
            key_pages=pool.key_pages,
            value_pages=pool.value_pages,
            block_table=torch.tensor(table, dtype=torch.int32, device=device).reshape(B, widest),
            context_lens=torch.tensor(context, dtype=torch.int32, device=device),
            page_size=pool.page_size,
            mask=mask,
            host_block_rows=table,
            host_context_lens=context,
        )

    def add_request(self, prefilled: TinyKvCache, id: int):
        if id >= self.max_active_requests:
            raise ValueError(f"Request id {id} is out of range")
        if isinstance(prefilled, TinyKvFullCache) and prefilled.key_values is not None:
            B, H, _, D = prefilled.key_values[0].shape
            assert B == 1
            if self.HD is None:
                self.HD = (H, D)
            else:
                assert self.HD == (H, D)
        self.kv_caches[id] = prefilled

    def remove_request(self, id: int):
        if self.kv_caches[id] is None:
            raise ValueError(f"Request id {id} is not in the cache")
        self.kv_caches[id].release()
        self.kv_caches[id] = None
