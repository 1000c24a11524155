"""Run ONLY through `pytest tests/moe_checkpoint_case.py -p refsol_oracle_plugin` (tests/test_loader_cpu.py does that): a
Qwen3-MoE checkpoint directory -> the product's loader (through the facade's `mlx_lm.load`) -> `Qwen3ModelWeek3` with Moe blocks
on the oracle-backed C ABI, against the facade's `mlx_lm` MoE model (fp32 torch, `mlx_lm.models.qwen3_moe`) on the same
tensors, in the pattern and with the tolerance of the reference's model tests (tests_refsol/test_week_3_day_1.py:150-195:
log-probabilities, rtol 0.1 / atol 2.0) and, tighter, on the greedy ids."""

import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def test_week3_model_on_a_loaded_moe_checkpoint(tmp_path):
    import mlx.core as mx
    from mlx_lm import load
    from tiny_llm_ref import Moe, Qwen3ModelWeek3

    from checkpoint_fixture import MOE_CFG_OVERRIDES, make_moe_weights, write_checkpoint
    from helpers import TINY_CFG

    cfg = dict(TINY_CFG, **MOE_CFG_OVERRIDES)
    ckpt = write_checkpoint(tmp_path / "moe", cfg, make_moe_weights(cfg, seed=5), vocab_words=[f"w{i}" for i in range(200)])
    mlx_model, tokenizer = load(str(ckpt))
    assert mlx_model.args.num_experts == 4 and mlx_model.args.mlp_only_layers == [0]
    model = Qwen3ModelWeek3(mlx_model, page_size=16)
    kinds = [type(layer.mlp).__name__ for layer in model.layers_inner]
    assert kinds == ["Qwen3MLP", "Moe", "Moe"], kinds
    assert isinstance(model.layers_inner[1].mlp, Moe)

    mx.random.seed(0)
    inputs = mx.random.randint(0, tokenizer.vocab_size, (1, 6))
    ref = mlx_model(inputs).float()
    ref = ref - mx.logsumexp(ref, axis=-1, keepdims=True)
    cache = model.create_kv_cache()
    agree = 0
    for offset in range(6):
        out = model(inputs=inputs[:, offset:offset + 1], offset=offset, cache=cache).float()
        out = out - mx.logsumexp(out, axis=-1, keepdims=True)
        want = ref[:, offset:offset + 1, :]
        np.testing.assert_allclose(np.array(out), np.array(want), rtol=0.1, atol=2.0)
        # far inside the reference's band: two bf16 pipelines over the same W4 tensors
        assert float((out - want).abs().max()) < 0.1, float((out - want).abs().max())
        agree += int(out.argmax(-1).item() == want.argmax(-1).item())
        print(f"offset {offset}: max abs log-prob difference {float((out - want).abs().max()):.4f}")
    assert agree >= 5, agree
    for layer_cache in cache:
        layer_cache.release()
