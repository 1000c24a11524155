"""Model dispatch (reference: src/tiny_llm_ref/models.py:8-18).  With TINY_LLM_FUSED_ENGINE=1 the Week-2 / Week-3 entry (no
``--week2-checkpoint``) is the fused decode engine behind the same call surface (engine_model.py)."""

from model_names import shortcut_name_to_full_name

from .qwen3_week1 import Qwen3ModelWeek1
from .qwen3_week2 import Qwen3ModelWeek2
from .qwen3_week3 import Qwen3ModelWeek3

_BY_WEEK = {1: Qwen3ModelWeek1, 2: Qwen3ModelWeek2, 3: Qwen3ModelWeek3}


def dispatch_model(model_name: str, mlx_model, week: int, **kwargs):
    full = shortcut_name_to_full_name(model_name)
    cls = _BY_WEEK.get(week)
    if cls is None or not full.startswith("Qwen/Qwen3"):
        raise ValueError(f"{full} for week {week} not supported")
    if week in (2, 3) and not kwargs.get("checkpoint"):
        from .engine_model import Qwen3ModelFused, fused_engine_requested

        if fused_engine_requested():  # TINY_LLM_FUSED_ENGINE=1: the same call surface on the fused decode engine
            return Qwen3ModelFused(mlx_model, **kwargs)
    return cls(mlx_model, **kwargs)
