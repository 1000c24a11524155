"""Greedy generation loops (reference: src/tiny_llm_ref/generate.py:16-81)."""

from typing import Callable

import torch


def _default_device(device=None) -> str:
    """The extension is GPU-only; without a GPU (CPU test tier, schedule-only runs) tensors stay on the host."""
    return device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")



def _release_kv_cache(kv_cache) -> None:
    if kv_cache is None:
        return
    for layer_cache in kv_cache:
        layer_cache.release()


def _log_softmax_last(logits: torch.Tensor) -> torch.Tensor:
    """Log-probabilities of the last position IN THE LOGITS' OWN DTYPE, as the reference forms them
    (generate.py:24-27,56-58: ``logits - mx.logsumexp(logits, keepdims=True)`` on the bf16 model output): a wider type here would
    break ties of the rounded values differently from the reference's argmax."""
    last = logits[:, -1, :]
    return last - torch.logsumexp(last, dim=-1, keepdim=True)


def simple_generate(model, tokenizer, prompt: str, sampler: Callable[[torch.Tensor], torch.Tensor] | None,
                    device: str | None = None, max_new_tokens: int | None = None) -> str:
    """Week-1 loop: the whole context is re-run for every token (no KV cache)."""
    tokens = torch.tensor(tokenizer.encode(prompt, add_special_tokens=False), dtype=torch.int32, device=_default_device(device))
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


def simple_generate_with_kv_cache(model, tokenizer, prompt: str, device: str | None = None,
                                  max_new_tokens: int | None = None) -> str:
    """Week-2/3 loop: one prefill call, then one token per call, one host sync per token."""
    kv_cache = model.create_kv_cache()
    try:
        tokens = torch.tensor(tokenizer.encode(prompt, add_special_tokens=False), dtype=torch.int32, device=_default_device(device))
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


def speculative_generate(draft_model, model, draft_tokenizer, tokenizer, prompt: str, proposal_length: int = 4,
                         device: str | None = None) -> str:
    """Greedy speculative decoding (reference: src/tiny_llm_ref/generate.py:84-322, tests_refsol/test_week_3_day_7.py).

    The draft model proposes up to ``proposal_length`` tokens; the target scores the pending token plus the proposals in ONE
    call (``logits_to_keep`` = number of positions, i.e. at most proposal_length + 1 query rows through the paged decode
    kernel) and keeps the longest prefix it would have produced itself; both KV caches are rewound to that point.  The text
    is exactly what target-only greedy decoding yields.  Error behaviour follows the reference: all tokenizer /
    argument checks run before any model call, caches are released on every exit path.
    """
    if not isinstance(proposal_length, int) or isinstance(proposal_length, bool) or proposal_length < 0:
        raise ValueError("proposal_length must be a non-negative integer")

    def encode(tok):
        return [int(t) for t in tok.encode(prompt, add_special_tokens=False)]

    def eos_set(tok):
        many = getattr(tok, "eos_token_ids", None)
        return {int(t) for t in (many if many is not None else {tok.eos_token_id})}

    prompt_ids = encode(tokenizer)
    if not prompt_ids:
        raise ValueError("prompt must encode to at least one token")
    if prompt_ids != encode(draft_tokenizer):
        raise ValueError("draft and target tokenizers encode the prompt differently")
    stop_ids = eos_set(tokenizer)
    draft_stop_ids = eos_set(draft_tokenizer)
    if stop_ids != draft_stop_ids:
        raise ValueError("draft and target tokenizers use different EOS token ids")
    vocab_of, draft_vocab_of = getattr(tokenizer, "get_vocab", None), getattr(draft_tokenizer, "get_vocab", None)
    if not callable(vocab_of) or not callable(draft_vocab_of):
        raise ValueError("draft and target tokenizers must expose comparable vocabularies")
    if vocab_of() != draft_vocab_of():
        raise ValueError("draft and target tokenizers use different token ids")

    detok = tokenizer.detokenizer
    detok.reset()

    def greedy(net, ids, offset, cache, keep=1):
        """argmax of the last `keep` positions of net(ids) appended at `offset`."""
        y = torch.tensor(ids, dtype=torch.int32, device=_default_device(device))
        logits = net(y[None], offset, cache, logits_to_keep=keep)[:, -keep:, :].to(torch.float32)
        return [int(t) for t in torch.argmax(logits, dim=-1).reshape(-1).tolist()]

    def rewind(cache, n):
        if n:
            for layer in cache:
                layer.rewind(n)

    def emit(ids):
        for t in ids:
            detok.add_token(t)
        if ids:
            print(f"+{len(ids)} " + detok.text.replace("\n", " ")[-80:])

    def finish():
        done = getattr(detok, "finalize", None)
        if callable(done):
            done()
        print(detok.text)
        return detok.text

    target_cache = model.create_kv_cache()
    draft_cache = None

    def target_only(token, offset):
        while token not in stop_ids:
            emit([token])
            token = greedy(model, [token], offset, target_cache)[0]
            offset += 1
        return finish()

    try:
        token = greedy(model, prompt_ids, 0, target_cache)[0]
        offset = len(prompt_ids)
        if token in stop_ids:
            return finish()
        if proposal_length == 0:
            return target_only(token, offset)

        draft_cache = draft_model.create_kv_cache()
        draft_first = greedy(draft_model, prompt_ids, 0, draft_cache)[0]
        draft_offset = len(prompt_ids)
        if draft_first in draft_stop_ids:
            return target_only(token, offset)

        def draft_run(last, budget):
            """Feed `last`, then the draft's own outputs, for at most `budget` steps; stops after proposing an EOS."""
            nonlocal draft_offset
            out = []
            for _ in range(budget):
                last = greedy(draft_model, [last], draft_offset, draft_cache)[0]
                draft_offset += 1
                out.append(last)
                if last in draft_stop_ids:
                    break
            return out

        while True:
            proposals = draft_run(token, proposal_length)
            fed = [token] + proposals                      # rows of the verification call
            predicted = greedy(model, fed, offset, target_cache, keep=len(fed))
            offset += len(fed)
            own = [token] + predicted[:-1]                 # what the target alone would have fed at each row
            cut, stopped = None, False
            for i, (mine, given) in enumerate(zip(own, fed)):
                if mine != given:
                    cut = i
                    break
                if mine in stop_ids:
                    cut, stopped = i, True
                    break
            if cut is not None:
                emit(own[:cut])
                rewind(target_cache, len(fed) - cut)
                rewind(draft_cache, len(proposals) - cut)
                offset -= len(fed) - cut
                draft_offset -= len(proposals) - cut
                assert offset == draft_offset
                if stopped or own[cut] in stop_ids:
                    return finish()
                token = own[cut]
                continue
            emit(own)
            bonus = predicted[-1]
            if bonus in stop_ids:
                return finish()
            draft_run(fed[-1], 1)                          # the draft has not seen its last proposal yet: catch up
            token = bonus
            assert offset == draft_offset
    finally:
        _release_kv_cache(draft_cache)
        _release_kv_cache(target_cache)
