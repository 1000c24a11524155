"""`mlx_lm` name: `load` reads an MLX-format 4-bit checkpoint directory through the product's loader
(tiny_llm_hip/loader.py; reference main.py:96-190 calls mlx_lm.load(name) -> (model, tokenizer))."""


def load(name_or_path, *args, **kwargs):
    from tiny_llm_hip.loader import load as _load

    return _load(name_or_path)
