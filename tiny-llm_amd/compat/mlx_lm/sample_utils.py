"""`mlx_lm.sample_utils.make_sampler(temp, top_p=, top_k=)` (reference main.py:186-188) -> the product's sampler over
log-probabilities (tiny_llm_hip/sampler.py, reference src/tiny_llm_ref/sampler.py:5-25)."""
from tiny_llm_hip.sampler import make_sampler as _make_sampler


def make_sampler(temp: float = 0.0, top_p: float = 0.0, min_p: float = 0.0, min_tokens_to_keep: int = 1, top_k: int = 0, **_):
    return _make_sampler(temp, top_p if top_p else None, top_k if top_k else None)
