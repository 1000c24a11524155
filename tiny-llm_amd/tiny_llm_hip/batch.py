"""Chunked prefill + continuous batching scheduler (reference: src/tiny_llm_ref/batch.py)."""

from datetime import datetime

import torch

from .kv_cache import BatchingKvCache


def _step(model, y, offsets, kv_cache):
    """Greedy next token for every row of ``y`` [B, L]."""
    logits = model(y, offsets, kv_cache, logits_to_keep=1)[:, -1, :].to(torch.float32)
    return torch.argmax(logits - torch.logsumexp(logits, dim=-1, keepdim=True), dim=-1)


class Request:
    """One prompt moving through chunked prefill and then decode (reference batch.py:16-96)."""

    def __init__(self, model, tokenizer, prompt: str, prefill_max_step: int = 128, prompt_idx: int = 0,
                 max_seq_len: int | None = None, device: str | None = None):
        self.prompt = prompt
        self.model = model
        self.detokenizer = tokenizer.detokenizer.__class__(tokenizer._tokenizer)
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.prefill_tokens = torch.tensor(
            tokenizer.encode(prompt, add_special_tokens=False), dtype=torch.int32, device=self.device)
        if max_seq_len is not None and self.prefill_tokens.numel() > max_seq_len:
            raise ValueError(
                f"Prompt has {self.prefill_tokens.numel()} tokens, which exceeds max_seq_len={max_seq_len}")
        self.kv_cache = model.create_kv_cache()
        self.prefill_max_step = prefill_max_step
        self.max_seq_len = max_seq_len
        self.is_done = False
        self.is_prefill_done = False
        self.finish_reason = None
        self.eos_token_id = tokenizer.eos_token_id
        self.next_token = None
        self.offset = 0
        self.prompt_idx = prompt_idx

    def try_prefill(self):
        """Push at most ``prefill_max_step`` prompt tokens through the model."""
        if self.is_prefill_done:
            raise ValueError("prefill called after done")
        total = self.prefill_tokens.numel()
        chunk = min(self.prefill_max_step, total - self.offset)
        token = _step(self.model, self.prefill_tokens[self.offset : self.offset + chunk][None], [self.offset],
                      self.kv_cache)
        self.offset += chunk
        for layer_cache in self.kv_cache:
            layer_cache.materialize()
        if self.offset == total:
            self.is_prefill_done = True
            if self.max_seq_len is not None and self.offset >= self.max_seq_len:
                self.is_done = True
                self.finish_reason = "max seq len"
            else:
                self.decode_done(int(token.item()), False)

    def decode_done(self, token, update_offset=True):
        if self.is_done:
            raise ValueError("decode called after done")
        if token == self.eos_token_id:
            self.is_done = True
            self.finish_reason = "EOS"
            return
        self.detokenizer.add_token(token)
        self.next_token = token
        if update_offset:
            self.offset += 1

    def text(self):
        return self.detokenizer.text

    def reaches_max_seq_len(self, max_seq_len: int) -> bool:
        # next_token is already emitted but not yet in the KV cache: it sits at position offset
        return self.next_token is not None and self.offset + 1 >= max_seq_len


def _print_progress(requests, pending, queue_size: int, tick: int, start_time: datetime) -> None:
    spinner = "⠋⠙⠹⠸⠼⠴⠦⠧⠇⠏"[tick % 10]
    print(f"  --- {datetime.now() - start_time}")
    for slot, request in enumerate(requests):
        if request is None:
            print(f"  Decode #{slot}: idle", flush=True)
        else:
            tail = request.text()[-80:].replace("\n", " ")
            print(f"{spinner} Decode [req {request.prompt_idx}, {request.offset}]: {tail}", flush=True)
    if pending is None:
        print(f"  Prefill: idle, {queue_size} requests in queue", flush=True)
    elif pending.is_prefill_done:
        print(f"  Prefill [req {pending.prompt_idx}]: done, waiting for slot, {queue_size} requests in queue",
              flush=True)
    else:
        total = pending.prefill_tokens.numel()
        print(f"{spinner} Prefill [req {pending.prompt_idx}]: {pending.offset / total * 100:.2f}% "
              f"({total - pending.offset} remaining tokens)", flush=True)


def batch_generate(model, tokenizer, prompts: list[str], max_seq_len=512, batch_size=5, prefill_step=128):
    """Serve ``prompts`` with at most one prefill chunk and one batched decode step per loop turn.
    Returns [(prompt_idx, text)] in completion order; every cache is released even on failure."""
    if max_seq_len <= 0:
        raise ValueError("max_seq_len must be positive")
    if batch_size <= 0:
        raise ValueError("batch_size must be positive")
    if prefill_step <= 0:
        raise ValueError("prefill_step must be positive")

    queue = list(prompts)
    slots: list[Request | None] = [None] * batch_size
    kv_cache = [BatchingKvCache(max_active_requests=batch_size, max_seq_len=max_seq_len)
                for _ in range(model.num_hidden_layers)]
    finished = []
    pending: Request | None = None
    issued = 0
    tick = 0
    started = datetime.now()

    try:
        while queue or pending is not None or any(r is not None for r in slots):
            if queue and pending is None:
                pending = Request(model, tokenizer, queue.pop(0), prefill_step, issued, max_seq_len=max_seq_len)
                issued += 1

            if pending is not None:
                progressed = False
                if not pending.is_prefill_done:
                    pending.try_prefill()
                    progressed = True
                if pending.is_prefill_done:
                    if pending.is_done or pending.reaches_max_seq_len(max_seq_len):
                        text = pending.text()
                        for request_cache in pending.kv_cache:
                            request_cache.release()
                        finished.append((pending.prompt_idx, text))
                        pending = None
                        progressed = True
                    else:
                        free = next((i for i, r in enumerate(slots) if r is None), None)
                        if free is not None:
                            for request_cache, batch_cache in zip(pending.kv_cache, kv_cache):
                                batch_cache.add_request(request_cache, free)
                            slots[free] = pending
                            pending = None
                            progressed = True
                if progressed:
                    _print_progress(slots, pending, len(queue), tick, started)
                    tick += 1

            if any(r is not None for r in slots):
                feed = [0 if r is None else r.next_token for r in slots]
                offsets = [0 if r is None else r.offset for r in slots]
                device = next(r.device for r in slots if r is not None)
                decoded = _step(model, torch.tensor(feed, dtype=torch.int32, device=device).reshape(-1, 1),
                                offsets, kv_cache).tolist()
                for slot, request in enumerate(slots):
                    if request is None:
                        continue
                    request.decode_done(decoded[slot])
                    reason = request.finish_reason if request.is_done else (
                        "max seq len" if request.reaches_max_seq_len(max_seq_len) else None)
                    if reason is not None:
                        print(f"Removing request {slot} due to {reason}", flush=True)
                        text = request.text()
                        for layer_cache in kv_cache:
                            layer_cache.remove_request(slot)
                        finished.append((request.prompt_idx, text))
                        slots[slot] = None
                _print_progress(slots, pending, len(queue), tick, started)
                tick += 1
    finally:
        # a cache can be referenced by the pending request AND by a half-filled batch slot: release once
        live = {}
        if pending is not None:
            for request_cache in pending.kv_cache:
                live[id(request_cache)] = request_cache
        for batch_cache in kv_cache:
            for request_cache in batch_cache.kv_caches:
                if request_cache is not None:
                    live[id(request_cache)] = request_cache
            batch_cache.kv_caches = [None] * batch_cache.max_active_requests
        for request_cache in live.values():
            request_cache.release()
    return finished
