"""GPU tier (collected last: written after the last GPU run of round 2): the engine-backed model
(tiny_llm_hip/engine_model.py) driven call for call like the reference's single-request bench loop
(benches/bench.py:run_one_request_week2, 277-312).  The logits it hands back at every step -- teacher-forced on the ids it
produced -- are held against the bf16 oracle and the float64 truth like every other model-level test
(helpers.check_against_truth: HIP error <= 1.5 x the oracle's own error); the adapter logic itself is covered on the CPU
(tests/test_engine_model_cpu.py)."""

import numpy as np
import pytest
import torch

from helpers import TINY_CFG, check_against_truth, to_mlx_shaped
from oracle import tiny_oracle as O

# First device run: profiles/r02_labs/zz_gpu_tests_first_device_run.log (all passed).
pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize("prompt_len,new_tokens,keep", [(23, 9, None), (1, 5, 1), (150, 6, 1)])
def test_reference_bench_loop_on_the_engine_backed_model(prompt_len, new_tokens, keep, monkeypatch):
    import tiny_llm_hip.models as models
    from tiny_llm_hip.engine_model import Qwen3ModelFused

    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    monkeypatch.setenv("TINY_LLM_FUSED_ENGINE", "1")
    monkeypatch.setenv("TINY_LLM_FUSED_MAX_CONTEXT", "512")
    model = models.dispatch_model("qwen3-4b", to_mlx_shaped(TINY_CFG, w), week=3, page_size=16)
    assert isinstance(model, Qwen3ModelFused)
    prompt = [int(t) for t in np.random.default_rng(prompt_len).integers(1, TINY_CFG["vocab_size"], size=prompt_len)]
    try:
        for _ in range(2):  # the second request reuses the slot (and the captured graphs)
            cache = model.create_kv_cache()
            rows, ids = [], []
            try:
                context = torch.tensor(prompt, dtype=torch.int32, device="cuda")
                logits = model(context[None, :], 0, cache, logits_to_keep=keep)
                offset = len(prompt)
                for step in range(new_tokens):
                    assert tuple(logits.shape) == (1, 1, TINY_CFG["vocab_size"])
                    rows.append(logits[0, -1].float().cpu().numpy())
                    token = torch.argmax(logits[:, -1, :], dim=-1)
                    ids.append(int(token))
                    if step + 1 < new_tokens:
                        logits = model(token.to(torch.int32)[None, :], offset, cache, logits_to_keep=1)
                        offset += 1
                assert cache[0].offset == len(prompt) + new_tokens - 1
            finally:
                for layer_cache in cache:
                    layer_cache.release()
            oracle, truth = O.OracleQwen3(TINY_CFG, w), O.TruthQwen3(TINY_CFG, w)
            want_o, want_t = [oracle.forward(prompt)[0, -1]], [truth.forward(prompt)[0, -1]]
            for tok in ids[:-1]:  # teacher-forced on the ids the engine produced
                want_o.append(oracle.forward([tok])[0, -1])
                want_t.append(truth.forward([tok])[0, -1])
            check_against_truth(np.stack(rows), np.stack(want_o), np.stack(want_t),
                                what=f"engine-backed model, {prompt_len}-token prompt, {new_tokens} steps")
        assert model.engine.stats()["pages_in_use"] == 0
    finally:
        model.close()
