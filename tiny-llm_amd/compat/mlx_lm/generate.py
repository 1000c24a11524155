"""`mlx_lm.generate.stream_generate(model, tokenizer, prompt, sampler=None, max_tokens=256)` (reference main.py:90,189-190:
the `--solution mlx` path): KV-cached generation on the facade's `mlx_lm` model, one response object per token with the
newly decoded text in `.text`."""

from types import SimpleNamespace

import torch

from .models.cache import make_prompt_cache


def stream_generate(model, tokenizer, prompt, max_tokens: int = 256, sampler=None, **_):
    ids = tokenizer.encode(prompt) if isinstance(prompt, str) else list(prompt)
    device = model.model.norm.weight.device
    cache = make_prompt_cache(model)
    detok = tokenizer.detokenizer
    detok.reset()
    tokens = torch.tensor([ids], dtype=torch.int32, device=device)
    for n in range(max_tokens):
        logits = model(tokens, cache=cache)[:, -1, :].float()
        logprobs = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        token = int((sampler(logprobs) if sampler is not None else torch.argmax(logprobs, dim=-1)).reshape(-1)[0])
        if token in tokenizer.eos_token_ids:
            break
        detok.add_token(token)
        yield SimpleNamespace(text=detok.last_segment, token=token, generation_tokens=n + 1)
        tokens = torch.tensor([[token]], dtype=torch.int32, device=device)
    detok.finalize()
    if detok.last_segment:
        yield SimpleNamespace(text=detok.last_segment, token=None, generation_tokens=None)


def generate(model, tokenizer, prompt, max_tokens: int = 256, sampler=None, verbose: bool = False, **kwargs) -> str:
    return "".join(r.text for r in stream_generate(model, tokenizer, prompt, max_tokens=max_tokens, sampler=sampler, **kwargs))
