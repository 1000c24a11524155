"""`tiny_llm_ref` import name -> the product package `tiny_llm_hip` (same public names, reference
src/tiny_llm_ref/__init__.py:1-20), with every submodule name aliased so that `importlib.import_module("tiny_llm_ref.x")`,
`from tiny_llm_ref.attention import ...` and monkeypatching of module attributes hit the product's modules."""
import importlib as _importlib
import sys as _sys

import tiny_llm_hip as _impl
from tiny_llm_hip import *  # noqa: F401,F403

# names the product package adds to the reference's surface: a test that does `from mlx_lm import load` and then
# `from tiny_llm_ref import *` (tests_refsol/test_week_3_day_1.py:6-8) must keep mlx_lm's `load`
for _extra in ("load", "load_weights"):
    globals().pop(_extra, None)

for _name in ("attention", "basics", "batch", "embedding", "generate", "kv_cache", "layer_norm", "models", "moe",
              "paged_kv_cache", "positional_encoding", "quantize", "qwen3_week1", "qwen3_week2", "qwen3_week3", "sampler",
              "week2_kernels", "loader", "engine"):
    _mod = _importlib.import_module(f"tiny_llm_hip.{_name}")
    _sys.modules[f"{__name__}.{_name}"] = _mod
    globals()[_name] = _mod
