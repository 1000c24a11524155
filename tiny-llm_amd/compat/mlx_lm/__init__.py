"""`mlx_lm` name: `load` reads an MLX-format 4-bit checkpoint directory (or a repo id in the local Hugging Face cache)
through the product's loader (tiny_llm_hip/loader.py; reference main.py:96-190 calls mlx_lm.load(name) -> (model,
tokenizer)) and hands the weight tree back as a facade `mlx_lm.models.qwen3.Model`: the same attribute tree the course
models read, plus the forward pass the reference's tests and `benches/bench.py --solution mlx` use as their oracle."""

__version__ = "0.31.3+torch-facade"


def load(name_or_path, *args, **kwargs):
    import mlx.core as mx
    from tiny_llm_hip.loader import load as _load

    from .models.qwen3 import Model

    tree, tokenizer = _load(name_or_path, device=str(mx._dev()))
    return Model.from_checkpoint(tree), tokenizer
