"""Token samplers over log-probabilities [B, V] (reference: src/tiny_llm_ref/sampler.py:5-25)."""

import torch


def make_sampler(temp: float, top_p: float | None, top_k: int | None):
    def sample(logprobs: torch.Tensor) -> torch.Tensor:
        if temp == 0:
            return torch.argmax(logprobs, dim=-1)
        scores = logprobs.clone()
        if top_k is not None and top_k > 0:
            kth = torch.topk(scores, top_k, dim=-1).values[..., -1:]
            scores = torch.where(scores < kth, torch.full_like(scores, float("-inf")), scores)
        if top_p is not None and top_p > 0:
            ordered, order = torch.sort(scores, dim=-1, descending=True)
            probs = torch.exp(ordered)
            before = torch.cumsum(probs, dim=-1) - probs
            ordered = torch.where(before < top_p, ordered, torch.full_like(ordered, float("-inf")))
            scores = torch.full_like(scores, float("-inf")).scatter(-1, order, ordered)
        dist = torch.softmax((scores / temp).to(torch.float32), dim=-1)
        return torch.multinomial(dist, 1).squeeze(-1)

    return sample
