"""The fused decode engine behind the course model's CALL SURFACE, so that harness code written against
``Qwen3ModelWeek2/3`` -- in particular the reference's own ``benches/bench.py:run_one_request_week2`` (277-312) and
``main.py`` / ``simple_generate_with_kv_cache`` -- drives the fused path without being changed:

    model = Qwen3ModelFused(mlx_model)              # or: TINY_LLM_FUSED_ENGINE=1 + models.dispatch_model(..., week=2 | 3)
    cache = model.create_kv_cache()                 # one handle per layer, all naming the request's engine slot
    logits = model(tokens[None, :], 0, cache)       # prefill: K/V appended, [1, 1, vocab] = logits of the LAST row
    logits = model(token[None, :], offset, cache)   # one decode step (5 fused kernels per layer, replayed hipGraph)
    for c in cache: c.release()

What is NOT the course model's behaviour, by construction of the engine (tinyllm_engine.h): only the last position's logits
exist (``logits_to_keep`` other than 1 still yields one row -- every caller in the reference's harness reads ``[:, -1, :]``),
one request per cache list (rows of ``inputs`` > 1 are refused: continuous batching goes through
``tiny_llm_hip.engine.batch_generate_ids`` / ``benches/serving.py``), K/V live in the engine's page pool (the cache handles
carry no tensors), and tokens must arrive in order (``offset`` must equal the slot's context length; ``rewind`` steps back).
"""

from __future__ import annotations

import os
from typing import Any, Callable

import torch

__all__ = ["Qwen3ModelFused", "fused_engine_requested"]


def fused_engine_requested() -> bool:
    """TINY_LLM_FUSED_ENGINE=1: models.dispatch_model(week=2 | 3) hands out the engine-backed model."""
    return os.environ.get("TINY_LLM_FUSED_ENGINE", "0") not in ("", "0")


class _SlotHandle:
    """What ``create_kv_cache`` returns per layer.  All handles of one request share the slot; the first ``release`` gives
    the slot (and its pages) back, the others are no-ops -- the harness releases every layer's cache in a loop.  Likewise
    ``rewind(n)`` (speculative decoding rewinds every layer's cache, reference generate.py:84-322) acts once per round: the
    slot is ONE sequence across all layers, so only layer 0's handle forwards it."""

    def __init__(self, owner: "Qwen3ModelFused", state: dict, layer: int):
        self._owner, self._state, self._layer = owner, state, layer

    @property
    def slot(self) -> int:
        return self._state["slot"]

    @property
    def offset(self) -> int:
        return self._owner.engine.context_len(self._state["slot"]) if self._state["live"] else 0

    def rewind(self, n: int) -> None:
        if self._layer == 0 and self._state["live"] and n > 0:
            self._owner.engine.rewind(self._state["slot"], n)

    def release(self) -> None:
        self._owner._release(self._state)


class Qwen3ModelFused:
    def __init__(self, mlx_model: Any, page_size: int = 128, enable_paged_attention: bool = True, *,
                 max_context: int | None = None, max_prefill_rows: int = 2048,
                 engine_factory: Callable[..., Any] | None = None, **_ignored):
        args = mlx_model.args
        self.num_hidden_layers = args.num_hidden_layers
        self.hidden_size = args.hidden_size
        self.vocab_size = args.vocab_size
        self.page_size = page_size
        self.precision = torch.bfloat16
        if max_context is None:
            max_context = int(os.environ.get("TINY_LLM_FUSED_MAX_CONTEXT", min(getattr(args, "max_position_embeddings", 40960), 40960)))
        pages_per_seq = (max_context + page_size - 1) // page_size + 1
        if engine_factory is None:
            from .engine import DecodeEngine as engine_factory  # the GPU-only extension is imported here, not at module import
        # ONE slot: a decode step advances every live slot below the batch size, so a second request would be stepped
        # along with the first; concurrent requests are the scheduler's business (engine.batch_generate_ids)
        self.engine = engine_factory(mlx_model, page_size=page_size, num_pages=pages_per_seq, max_batch=1,
                                     max_pages_per_seq=pages_per_seq, max_prefill_rows=max_prefill_rows)
        self.max_prefill_rows = max_prefill_rows
        self._free_slots = [0]
        self.page_pools = ()  # the engine owns the pages (benches/bench.py resets `page_pools` after its warm-up if present)

    # -- the course model's surface ----------------------------------------------------------------------------------
    def create_kv_cache(self) -> list[_SlotHandle]:
        if not self._free_slots:
            raise RuntimeError("Qwen3ModelFused: the previous request's cache has not been released (one request at a time)")
        state = {"slot": self._free_slots.pop(), "live": True}
        self.engine.begin(state["slot"])
        return [_SlotHandle(self, state, layer) for layer in range(self.num_hidden_layers)]

    def __call__(self, inputs: torch.Tensor, offset: Any, cache: list[_SlotHandle], mask: Any = None,
                 logits_to_keep: int | None = None) -> torch.Tensor:
        if inputs.dim() != 2 or inputs.shape[0] != 1:
            raise ValueError("Qwen3ModelFused: one request per call ([1, L] token ids); batches go through batch_generate_ids")
        if logits_to_keep not in (None, 1):
            raise ValueError("Qwen3ModelFused returns the logits of the LAST position only (logits_to_keep must be None or 1): "
                             "multi-position verification of speculative decoding goes through DecodeEngine.verify")
        if mask is not None and not (isinstance(mask, str) and mask == "causal"):
            raise ValueError("Qwen3ModelFused applies the causal mask itself; explicit masks are not supported")
        if not cache or not isinstance(cache[0], _SlotHandle) or not cache[0]._state["live"]:
            raise ValueError("Qwen3ModelFused: cache must come from this model's create_kv_cache() and not be released")
        state = cache[0]._state
        slot = state["slot"]
        start = int(offset if not isinstance(offset, torch.Tensor) else offset.reshape(-1)[0].item())
        have = self.engine.context_len(slot)
        if start != have:
            raise ValueError(f"Qwen3ModelFused: tokens must arrive in order (offset {start}, context so far {have})")
        tokens = [int(t) for t in inputs.reshape(-1).tolist()]
        if len(tokens) == 1 and have > 0:
            # one decode step: the token is the slot's pending input (the harness passes back the id it sampled)
            self.engine.set_token(slot, tokens[0])
            self.engine.decode(1, batch=1)
            row = self.engine.logits(1)
        else:
            self.engine.prefill(slot, tokens, chunk=self.max_prefill_rows)
            row = self.engine.logits(1)
        return row.reshape(1, 1, self.vocab_size)

    # -- slot bookkeeping --------------------------------------------------------------------------------------------
    def _release(self, state: dict) -> None:
        if state["live"]:
            state["live"] = False
            self.engine.release(state["slot"])
            self._free_slots.append(state["slot"])

    def close(self) -> None:
        if getattr(self, "engine", None) is not None and hasattr(self.engine, "close"):
            self.engine.close()
