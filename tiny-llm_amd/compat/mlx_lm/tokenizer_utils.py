"""`mlx_lm.tokenizer_utils.TokenizerWrapper` (imported by reference generate.py:2 and batch.py:2 for type annotations and
the detokenizer protocol) -> the product loader's wrapper around a Hugging Face tokenizer (tiny_llm_hip/loader.py)."""
from tiny_llm_hip.loader import TokenizerWrapper  # noqa: F401
