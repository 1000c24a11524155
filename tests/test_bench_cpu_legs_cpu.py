"""CPU tier: the bounded CPU legs of bench.py (the torch-CPU restatement leg end to end on a small model; the C port's prompt
walk stops at its wall-clock budget) -- a leg that overruns takes the driver's bench line with it, as one did in round 4."""

import random
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from helpers import TINY_CFG
from oracle import tiny_oracle as O
from test_torch_week2_baseline_cpu import _dense


def test_torch_leg_reports_a_rate_and_respects_its_budget(monkeypatch):
    import bench

    weights = O.make_qwen3_weights(TINY_CFG, seed=9, sigma=0.05)
    prompt = bench.build_prompt(random.Random(1), 40, TINY_CFG["vocab_size"])
    out = bench.torch_week2_leg(None, TINY_CFG, prompt, [3, 4, 5, 6], None, dense=_dense(weights))
    assert out["value"] > 0 and out["cores"] >= 1 and "NOT MLX" in out["label"] and "40-token prompt" in out["sample"]
    assert out["linear_storage"] in ("bfloat16", "float32") and set(out["gemv_probe_ms"]) == {"bfloat16", "float32"}
    monkeypatch.setattr(bench, "TORCH_BUDGET_S", 0.0)  # nothing fits: the prompt is cut to its 8-token floor and the line says so
    out = bench.torch_week2_leg(None, TINY_CFG, prompt, [3, 4], None, dense=_dense(weights))
    assert "8-token prompt (cut from 40" in out["sample"] and out["max_abs_logit_vs_gpu_first_decode_step"] is None


def test_prompt_walk_is_bounded_and_in_lock_step():
    import bench

    class Slow:
        def __init__(self):
            self.fed = []

        def step(self, t):
            self.fed.append(t)
            return t + 1, np.zeros(4)

    a, b = Slow(), Slow()
    fed, last = bench.walk_prompt_bounded([a, b], list(range(100)), budget_s=0.0)
    assert fed == 8 and a.fed == b.fed == list(range(8)) and last[0][0] == last[1][0] == 8, "an exhausted budget stops both at the 8-token floor"
    a = Slow()
    fed, last = bench.walk_prompt_bounded([a], list(range(20)), budget_s=60.0)
    assert fed == 20 and last[0][0] == 20


def test_time_box_interrupts_a_slow_leg():
    import time

    import bench

    with pytest.raises(TimeoutError):
        with bench.time_box(1):
            for _ in range(100):
                time.sleep(0.1)
    with bench.time_box(5):
        pass  # leaving the box disarms it
    time.sleep(0.01)
