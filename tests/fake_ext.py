"""TEST-ONLY stand-in for ``tiny_llm_ext_hip`` on machines without a GPU.

Host-side logic (page pools, request caches, batching metadata, the scheduler) lives in Python above the
extension; to exercise it in the CPU test tier the consumer modules' ``tiny_llm_ext_hip`` attribute is
monkeypatched with this object, which answers the two ops that logic touches with the numpy oracle.  The
product never imports this file and has no CPU path (its extension raises for host tensors)."""

import numpy as np
import torch

from oracle import tiny_oracle as O


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()


class FakeExt:
    calls: list

    def __init__(self):
        self.calls = []

    def paged_cache_update(self, pages, values, page_id, start, stream=None):
        self.calls.append(("paged_cache_update", int(page_id), int(start), int(values.shape[2])))
        if page_id < 0 or page_id >= pages.shape[0] or start < 0 or start + values.shape[2] > pages.shape[2]:
            raise RuntimeError("paged_cache_update: destination slice is outside page storage")
        pages[page_id, :, start:start + values.shape[2], :] = values[0]  # in place, like the extension
        return pages

    def paged_attention(self, query, key_pages, value_pages, block_table, context_lens, scale=1.0, is_causal=False, *,
                        num_kv_heads, num_heads, stream=None, max_context_hint=0):
        self.calls.append(("paged_attention", tuple(query.shape)))
        dtype = "bf16" if query.dtype == torch.bfloat16 else "f32"
        out = O.paged_attention(_np(query), _np(key_pages), _np(value_pages), block_table.cpu().numpy(),
                                context_lens.cpu().numpy(), float(scale), bool(is_causal), num_kv_heads, num_heads, dtype)
        return torch.from_numpy(out).to(query.dtype)
