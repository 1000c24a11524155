"""Greedy generation loops (reference: src/tiny_llm_ref/generate.py:16-81)."""

from typing import Callable

import torch


def _release_kv_cache(kv_cache) -> None:
    if kv_cache is None:
        return
    for layer_cache in kv_cache:
        layer_cache.release()


def _log_softmax_last(logits: torch.Tensor) -> torch.Tensor:
    wide = logits[:, -1, :].to(torch.float32)
    return wide - torch.logsumexp(wide, dim=-1, keepdim=True)


def simple_generate(model, tokenizer, prompt: str, sampler: Callable[[torch.Tensor], torch.Tensor] | None,
                    device: str = "cuda", max_new_tokens: int | None = None) -> str:
    """Week-1 loop: the whole context is re-run for every token (no KV cache)."""
    tokens = torch.tensor(tokenizer.encode(prompt, add_special_tokens=False), dtype=torch.int32, device=device)
    detok = tokenizer.detokenizer
    detok.reset()
    produced = 0
    while max_new_tokens is None or produced < max_new_tokens:
        logprobs = _log_softmax_last(model(tokens[None]))
        token = torch.argmax(logprobs, dim=-1) if sampler is None else sampler(logprobs)
        token_id = int(token.item())
        tokens = torch.cat([tokens, token.to(tokens.dtype)])
        if token_id == tokenizer.eos_token_id:
            break
        detok.add_token(token_id)
        print(detok.last_segment, end="", flush=True)
        produced += 1
    return detok.text


def simple_generate_with_kv_cache(model, tokenizer, prompt: str, device: str = "cuda",
                                  max_new_tokens: int | None = None) -> str:
    """Week-2/3 loop: one prefill call, then one token per call, one host sync per token."""
    kv_cache = model.create_kv_cache()
    try:
        tokens = torch.tensor(tokenizer.encode(prompt, add_special_tokens=False), dtype=torch.int32, device=device)
        detok = tokenizer.detokenizer
        detok.reset()
        offset = 0
        produced = 0
        while max_new_tokens is None or produced < max_new_tokens:
            logprobs = _log_softmax_last(model(tokens[None], offset, kv_cache, logits_to_keep=1))
            token = torch.argmax(logprobs, dim=-1)
            token_id = int(token.item())
            if token_id == tokenizer.eos_token_id:
                break
            detok.add_token(token_id)
            print(detok.last_segment, end="", flush=True)
            offset += tokens.numel()  # first pass: prompt length; afterwards: 1
            tokens = token.to(torch.int32)
            produced += 1
        return detok.text
    finally:
        _release_kv_cache(kv_cache)
