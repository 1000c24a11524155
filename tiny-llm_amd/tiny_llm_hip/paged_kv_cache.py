"""Paged KV storage: layer-local page pool + per-request page tables
(reference: src/tiny_llm_ref/paged_kv_cache.py)."""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from ._ext import tiny_llm_ext_hip
from .kv_cache import TinyKvCache, materialize_tensors


@dataclass
class PagedKvMetadata:
    """What ``paged_attention`` needs for one layer call (reference paged_kv_cache.py:11-18).  The two
    ``host_*`` fields carry the integers the device tensors were built from, so the attention wrapper can
    validate them without reading the tensors back."""

    key_pages: torch.Tensor
    value_pages: torch.Tensor
    block_table: torch.Tensor
    context_lens: torch.Tensor
    page_size: int
    mask: torch.Tensor | str | None = None
    host_block_rows: list[list[int]] | None = field(default=None, compare=False)
    host_context_lens: list[int] | None = field(default=None, compare=False)


def _bytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


class TinyKvPagedPool:
    """Physical K and V page arrays [capacity, H, page_size, D] for ONE transformer layer, a free list, and
    the growth/reuse counters the serving benchmark reports (reference paged_kv_cache.py:21-242).

    Storage appears lazily with the first write and grows to ``max(4, pages_in_use, 2 * capacity)``,
    copying the pages that existed before (counted in ``copied_*_on_growth``).  On a 288 GB part the
    sensible deployment is ``reserve()``-ing the expected page count up front, after which the growth
    path (and its copies) never runs; the counters then stay at their reserved-start values."""

    def __init__(self, page_size: int = 128):
        assert page_size > 0
        self.page_size = page_size
        self._key_pages: torch.Tensor | None = None
        self._value_pages: torch.Tensor | None = None
        self.free_page_ids: list[int] = []
        self.used_page_ids: set[int] = set()
        self.num_allocated_pages = 0
        self.reused_page_allocations = 0
        self.storage_growths = 0
        self.copied_pages_on_growth = 0
        self.copied_bytes_on_growth = 0

    # -- views & sizes ---------------------------------------------------------------
    @property
    def num_pages(self) -> int:
        return self.num_allocated_pages

    @property
    def key_pages(self) -> torch.Tensor | None:
        return None if self._key_pages is None else self._key_pages[: self.num_pages]

    @property
    def value_pages(self) -> torch.Tensor | None:
        return None if self._value_pages is None else self._value_pages[: self.num_pages]

    @property
    def capacity(self) -> int:
        return 0 if self._key_pages is None else self._key_pages.shape[0]

    @property
    def num_free_pages(self) -> int:
        return len(self.free_page_ids)

    @property
    def storage_nbytes(self) -> int:
        if self._key_pages is None or self._value_pages is None:
            return 0
        return _bytes(self._key_pages) + _bytes(self._value_pages)

    # -- transactions -----------------------------------------------------------------
    _STATE_FIELDS = (
        "_key_pages", "_value_pages", "free_page_ids", "used_page_ids", "num_allocated_pages",
        "reused_page_allocations", "storage_growths", "copied_pages_on_growth", "copied_bytes_on_growth",
    )

    def _snapshot_state(self) -> tuple:
        return tuple(
            list(v) if isinstance(v, list) else set(v) if isinstance(v, set) else v
            for v in (getattr(self, name) for name in self._STATE_FIELDS)
        )

    def _restore_state(self, state: tuple) -> None:
        for name, value in zip(self._STATE_FIELDS, state):
            setattr(self, name, value)

    # -- validation -------------------------------------------------------------------
    def validate_page_chunk(self, key: torch.Tensor, value: torch.Tensor) -> None:
        if key.dim() != 4 or value.dim() != 4:
            raise ValueError("Paged K/V chunks must be 4D [1, H, S, D]")
        if key.shape != value.shape:
            raise ValueError("Paged K/V chunks must have the same shape")
        B, H, S, D = key.shape
        if B != 1:
            raise ValueError("Paged request cache only supports one request")
        if H <= 0 or D <= 0 or S <= 0:
            raise ValueError("Paged K/V chunks must have positive valid dimensions")
        if key.dtype != value.dtype or key.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("Paged K/V chunks must have the same float32 or bfloat16 dtype")
        if (self._key_pages is None) != (self._value_pages is None):
            raise ValueError("Paged K/V storage is incomplete")
        if self._key_pages is not None:
            if tuple(self._key_pages.shape[1:]) != (H, self.page_size, D):
                raise ValueError("Paged K/V chunks must match the existing page storage shape")
            if self._value_pages.shape != self._key_pages.shape:
                raise ValueError("Paged key and value storage must have the same shape")
            if self._key_pages.dtype != key.dtype or self._value_pages.dtype != value.dtype:
                raise ValueError("Paged K/V chunks must match the existing page storage dtype")

    # -- allocator --------------------------------------------------------------------
    def allocate_page(self) -> int:
        if self.free_page_ids:
            page_id = self.free_page_ids.pop()
            self.reused_page_allocations += 1
        else:
            page_id = self.num_allocated_pages
            self.num_allocated_pages += 1
        self.used_page_ids.add(page_id)
        return page_id

    def free_page(self, page_id: int) -> None:
        if page_id not in self.used_page_ids:
            raise ValueError(f"Page {page_id} is already free")
        self.used_page_ids.remove(page_id)  # id stays valid; stale bytes are masked by page tables
        self.free_page_ids.append(page_id)

    def read_page(self, page_id: int) -> tuple[torch.Tensor, torch.Tensor]:
        if self._key_pages is None or self._value_pages is None:
            raise ValueError(f"Page {page_id} has no storage")
        if page_id >= self.num_pages:
            raise ValueError(f"Page {page_id} is out of range")
        return self._key_pages[page_id : page_id + 1], self._value_pages[page_id : page_id + 1]

    def reserve(self, pages: int, heads: int, head_dim: int, dtype: torch.dtype, device) -> None:
        """Preallocate ``pages`` physical pages (MI355X deployment mode; not part of the reference API)."""
        if self._key_pages is not None and self.capacity >= pages:
            return
        shape = (pages, heads, self.page_size, head_dim)
        new_k = torch.zeros(shape, dtype=dtype, device=device)
        new_v = torch.zeros(shape, dtype=dtype, device=device)
        if self._key_pages is not None:
            live = self.num_pages
            new_k[:live] = self._key_pages[:live]
            new_v[:live] = self._value_pages[:live]
        self._key_pages, self._value_pages = new_k, new_v

    def _ensure_page_storage(self, key: torch.Tensor, value: torch.Tensor) -> None:
        _, H, _, D = key.shape
        if self._key_pages is not None and self._value_pages is not None:
            assert tuple(self._key_pages.shape[1:]) == (H, self.page_size, D)
            assert self._key_pages.dtype == key.dtype and self._value_pages.dtype == value.dtype
            if self.capacity >= self.num_pages:
                return
        grown = max(4, self.num_pages, self.capacity * 2)
        shape = (grown, H, self.page_size, D)
        new_k = torch.zeros(shape, dtype=key.dtype, device=key.device)
        new_v = torch.zeros(shape, dtype=value.dtype, device=value.device)
        self.storage_growths += 1
        if self._key_pages is not None and self._value_pages is not None:
            carried = self.num_pages - 1  # the page being written now has no old contents
            self.copied_pages_on_growth += carried
            self.copied_bytes_on_growth += _bytes(self._key_pages[:carried]) + _bytes(self._value_pages[:carried])
            new_k[:carried] = self._key_pages[:carried]
            new_v[:carried] = self._value_pages[:carried]
        self._key_pages, self._value_pages = new_k, new_v

    def reset(self) -> None:
        if self.used_page_ids:
            raise ValueError("Cannot reset a page pool with live requests")
        self._key_pages = None
        self._value_pages = None
        self.free_page_ids.clear()
        self.num_allocated_pages = 0
        self.reused_page_allocations = 0
        self.storage_growths = 0
        self.copied_pages_on_growth = 0
        self.copied_bytes_on_growth = 0

    def write_page_slice(self, page_id: int, start: int, key: torch.Tensor, value: torch.Tensor) -> None:
        """Write ``key``/``value`` [1, H, len, D] into slots [start, start+len) of one page (two scatter launches,
        reference paged_kv_cache.py:196-234)."""
        self.validate_page_chunk(key, value)
        if key.shape[2] > self.page_size:
            raise ValueError("Paged K/V writes cannot exceed one physical page")
        if page_id not in self.used_page_ids:
            raise ValueError(f"Page {page_id} is free")
        if page_id < 0 or page_id >= self.num_pages:
            raise ValueError(f"Page {page_id} is out of range")
        if start < 0 or start + key.shape[2] > self.page_size:
            raise ValueError("Paged K/V write is outside page storage")
        self._ensure_page_storage(key, value)
        self._key_pages = tiny_llm_ext_hip.paged_cache_update(self._key_pages, key.contiguous(), page_id, start)
        self._value_pages = tiny_llm_ext_hip.paged_cache_update(self._value_pages, value.contiguous(), page_id, start)


class TinyKvPagedCache(TinyKvCache):
    """Logical view of one request in one layer: ordered page ids, per-page fill, token count
    (reference paged_kv_cache.py:245-443)."""

    def __init__(self, pool: TinyKvPagedPool):
        self.pool = pool
        self.page_size = pool.page_size
        self.page_ids: list[int] = []
        self.page_lens: list[int] = []
        self.offset = 0
        self._cached_block_table: torch.Tensor | None = None
        self._cached_block_table_key: tuple[tuple[int, ...], int] | None = None

    @property
    def num_pages(self) -> int:
        return len(self.page_ids)

    @property
    def key_values(self) -> tuple[torch.Tensor, torch.Tensor] | None:
        return None if self.offset == 0 else self.gather_dense()

    def _snapshot_state(self) -> tuple:
        return (list(self.page_ids), list(self.page_lens), self.offset, self._cached_block_table,
                self._cached_block_table_key)

    def _restore_state(self, state: tuple) -> None:
        (self.page_ids, self.page_lens, self.offset, self._cached_block_table,
         self._cached_block_table_key) = state

    def validate_append(self, key: torch.Tensor, value: torch.Tensor) -> None:
        self.pool.validate_page_chunk(key, value)

    def _append_chunk(self, key: torch.Tensor, value: torch.Tensor) -> None:
        """Top up the tail page, then open new pages; on any error the pool and this cache roll back."""
        self.pool.validate_page_chunk(key, value)
        length = key.shape[2]
        mine = self._snapshot_state()
        theirs = self.pool._snapshot_state()
        try:
            done = 0
            if self.page_ids and self.page_lens[-1] < self.page_size:
                at = self.page_lens[-1]
                take = min(self.page_size - at, length)
                self.pool.write_page_slice(self.page_ids[-1], at, key[:, :, :take, :], value[:, :, :take, :])
                self.page_lens[-1] += take
                done = take
            while done < length:
                stop = min(done + self.page_size, length)
                page_id = self.pool.allocate_page()
                self.pool.write_page_slice(page_id, 0, key[:, :, done:stop, :], value[:, :, done:stop, :])
                self.page_ids.append(page_id)
                self.page_lens.append(stop - done)
                done = stop
            self.offset += length
        except Exception:
            self.pool._restore_state(theirs)
            self._restore_state(mine)
            raise

    def gather_dense(self) -> tuple[torch.Tensor, torch.Tensor]:
        """Dense [1, H, offset, D] copy (compatibility path; paged attention never calls it)."""
        assert self.offset > 0
        ks, vs = [], []
        for page_id, fill in zip(self.page_ids, self.page_lens):
            k, v = self.pool.read_page(page_id)
            ks.append(k[:, :, :fill, :])
            vs.append(v[:, :, :fill, :])
        if len(ks) == 1:
            return ks[0], vs[0]
        return torch.cat(ks, dim=2), torch.cat(vs, dim=2)

    def update_and_fetch(self, key, value, mask_length=None, mask=None):
        self._append_chunk(key, value)
        k, v = self.gather_dense()
        return k, v, self.offset, mask

    def block_table(self, max_pages: int | None = None) -> torch.Tensor:
        """[1, max_pages] int32, -1 padded; the same tensor object is returned until the page ids change."""
        width = self.num_pages if max_pages is None else max_pages
        assert width >= self.num_pages
        key = (tuple(self.page_ids), width)
        if self._cached_block_table is None or self._cached_block_table_key != key:
            device = self.pool.key_pages.device if self.pool.key_pages is not None else None
            row = self.page_ids + [-1] * (width - self.num_pages)
            self._cached_block_table = torch.tensor([row], dtype=torch.int32, device=device).reshape(1, width)
            self._cached_block_table_key = key
        return self._cached_block_table

    def context_lens(self) -> torch.Tensor:
        device = self.pool.key_pages.device if self.pool.key_pages is not None else None
        return torch.tensor([self.offset], dtype=torch.int32, device=device)

    def paged_metadata(self, max_pages: int | None = None, mask=None) -> PagedKvMetadata:
        assert self.pool.key_pages is not None and self.pool.value_pages is not None
        width = self.num_pages if max_pages is None else max_pages
        return PagedKvMetadata(
            key_pages=self.pool.key_pages,
            value_pages=self.pool.value_pages,
            block_table=self.block_table(max_pages=max_pages),
            context_lens=self.context_lens(),
            page_size=self.page_size,
            mask=mask,
            host_block_rows=[self.page_ids + [-1] * (width - self.num_pages)],
            host_context_lens=[self.offset],
        )

    def update_and_fetch_paged(self, key, value, mask_length=None, mask=None) -> PagedKvMetadata:
        self._append_chunk(key, value)
        return self.paged_metadata(mask=mask)

    def materialize(self):
        k, v = self.pool.key_pages, self.pool.value_pages
        if k is not None and v is not None:
            materialize_tensors(k, v)

    def rewind(self, n: int):
        assert 0 <= n <= self.offset
        keep = self.offset - n
        if n == 0:
            return
        if keep == 0:
            self.release()
            return
        pages_needed = (keep + self.page_size - 1) // self.page_size
        while len(self.page_ids) > pages_needed:
            self.pool.free_page(self.page_ids.pop())
            self.page_lens.pop()
        self.page_lens[-1] = keep - self.page_size * (pages_needed - 1)
        self.offset = keep

    def release(self):
        for page_id in self.page_ids:
            self.pool.free_page(page_id)
        self.page_ids.clear()
        self.page_lens.clear()
        self.offset = 0
