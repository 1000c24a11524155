"""Model-level cases in the pattern of the reference's checkpoint-dependent tests (which it skips without a downloaded
model): a checkpoint DIRECTORY -> the product's loader through the facade's `mlx_lm.load` -> the course model on the C ABI,
against the facade's `mlx_lm` model (fp32 torch restatement, shares no code with the numpy oracle) on the same tensors.

* Week 1 (tests_refsol/test_week_1_day_5.py:110-123), Week 2 incremental decode (test_week_2_day_6.py:124-148), Week 3
  staggered continuous batching (test_week_3_day_1.py:128-195), each with the reference's assertion (log-probabilities, rtol
  0.1, atol 2.5 / 2.0) AND a band that means something: the two pipelines round to bf16 at the same op boundaries, measured
  apart 0.02-0.05 on these checkpoints (oracle-backed run), asserted < 0.15; the greedy id may differ only where the oracle's own margin is
  inside that band.
* the Week-3 model on a loaded Qwen3-MoE checkpoint (router + stacked experts; reference qwen3_week3.py:258-272).

This file is not collected by `pytest tests/` (name).  It runs
  * on the MI355X from tests/test_zz_facade_models_gpu.py (the HIP kernels answer), and
  * in the build container through `pytest tests/facade_model_cases.py -p refsol_oracle_plugin` (the numpy oracle answers the
    C ABI), started by tests/test_loader_cpu.py -- so every line below is executed before it reaches the GPU box.
"""

import numpy as np

BAND = 0.15  # max |log-prob difference| between the course model on the C ABI and the facade's mlx_lm model


def _log_softmax(mx, x):
    x = x.float()
    return x - mx.logsumexp(x, axis=-1, keepdims=True)


def _check(mx, got, want, ref_atol):
    got, want = _log_softmax(mx, got), _log_softmax(mx, want)
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=0.1, atol=ref_atol)  # the reference's own assertion
    diff = float((got - want).abs().max())
    assert diff < BAND, diff
    top = want.argmax(-1)
    margin = want.max(-1).values - mx.take_along_axis(want, got.argmax(-1)[..., None], axis=-1)[..., 0]
    assert bool((margin <= 2 * BAND).all()), (got.argmax(-1).tolist(), top.tolist())  # same greedy id unless inside the band
    return diff


def _write(tmp_path, name, overrides, seed, moe=False):
    from checkpoint_fixture import make_moe_weights, write_checkpoint
    from helpers import TINY_CFG
    from oracle import tiny_oracle as O

    cfg = dict(TINY_CFG, **overrides)
    w = make_moe_weights(cfg, seed=seed) if moe else O.make_qwen3_weights(cfg, seed=seed, sigma=0.05)
    return write_checkpoint(tmp_path / name, cfg, w, vocab_words=[f"w{i}" for i in range(200)])


def case_week1_model(tmp_path):
    import mlx.core as mx
    from mlx_lm import load
    from tiny_llm_ref import Qwen3ModelWeek1

    mlx_model, tokenizer = load(str(_write(tmp_path, "w1", dict(), 31)))
    model = Qwen3ModelWeek1(mlx_model)
    worst = 0.0
    for iteration in range(3):
        inputs = (mx.arange(10, dtype=mx.int32) + iteration * 10).reshape(1, 10) % tokenizer.vocab_size
        worst = max(worst, _check(mx, model(inputs), mlx_model(inputs), 2.5))
    return worst


def case_week2_incremental_decode(tmp_path, checkpoint="split-k"):
    import mlx.core as mx
    from mlx_lm import load
    from tiny_llm_ref import Qwen3ModelWeek2

    mlx_model, tokenizer = load(str(_write(tmp_path, "w2", dict(tie_word_embeddings=False), 32)))
    model = Qwen3ModelWeek2(mlx_model, checkpoint=checkpoint)
    mx.random.seed(1)
    seq_len = 5
    inputs = mx.random.randint(0, tokenizer.vocab_size, (1, seq_len))
    ref = mlx_model(inputs)
    cache = model.create_kv_cache()
    worst = 0.0
    try:
        for offset in range(seq_len):
            out = model(inputs=inputs[:, offset:offset + 1], offset=offset, cache=cache)
            worst = max(worst, _check(mx, out, ref[:, offset:offset + 1, :], 2.5))
    finally:
        for layer_cache in cache:
            layer_cache.release()
    return worst


def case_week3_staggered_batching(tmp_path, moe=False):
    """Three requests that join two steps apart and leave when done (reference helper_test_task_3)."""
    import mlx.core as mx
    from mlx_lm import load
    from tiny_llm_ref import BatchingKvCache, Moe, Qwen3ModelWeek3

    from checkpoint_fixture import MOE_CFG_OVERRIDES

    overrides = dict(MOE_CFG_OVERRIDES) if moe else dict(num_hidden_layers=3)
    mlx_model, tokenizer = load(str(_write(tmp_path, "w3moe" if moe else "w3", overrides, 33, moe=moe)))
    model = Qwen3ModelWeek3(mlx_model, page_size=16)
    if moe:
        assert mlx_model.args.num_experts == 4 and mlx_model.args.mlp_only_layers == [0]
        assert [type(layer.mlp).__name__ for layer in model.layers_inner] == ["Qwen3MLP", "Moe", "Moe"]
        assert isinstance(model.layers_inner[1].mlp, Moe)
    seq_len, starts = 4, [0, 2, 4]
    mx.random.seed(2)
    inputs = mx.random.randint(0, tokenizer.vocab_size, (len(starts), seq_len))
    ref = mlx_model(inputs)
    cache = [BatchingKvCache(max_active_requests=len(starts), max_seq_len=64) for _ in range(model.num_hidden_layers)]
    per_request = {}
    worst = 0.0
    for step in range(seq_len + starts[-1]):
        index = [step - start for start in starts]
        for request_id, sidx in enumerate(index):
            if sidx == 0:
                per_request[request_id] = model.create_kv_cache()
                for c, own in zip(cache, per_request[request_id]):
                    c.add_request(own, request_id)
            elif sidx == seq_len:
                for c in cache:
                    c.remove_request(request_id)
        tokens = [int(inputs[r, s].item()) if 0 <= s < seq_len else 0 for r, s in enumerate(index)]
        offsets = [s if 0 <= s < seq_len else 0 for s in index]
        out = model(inputs=mx.array(tokens, dtype=mx.int32).reshape(-1, 1), offset=mx.array(offsets, dtype=mx.int32), cache=cache)
        for request_id, sidx in enumerate(index):
            if 0 <= sidx < seq_len:
                worst = max(worst, _check(mx, out[request_id, 0, :], ref[request_id, sidx, :], 2.0))
    for c in cache:
        c.remove_request(len(starts) - 1)
    for pool in model.page_pools:
        assert pool.num_free_pages == pool.num_pages  # every page came back
    return worst


# ---- collected only when this file is named on the pytest command line (CPU: with -p refsol_oracle_plugin) -------------------
def test_week1_model(tmp_path):
    print("week 1 model: max |log-prob difference|", case_week1_model(tmp_path))


def test_week2_incremental_decode(tmp_path):
    for checkpoint in ("kv-cache", "split-k"):
        print(f"week 2 {checkpoint}: max |log-prob difference|", case_week2_incremental_decode(tmp_path, checkpoint))


def test_week3_staggered_batching(tmp_path):
    print("week 3 batching: max |log-prob difference|", case_week3_staggered_batching(tmp_path))


def test_week3_staggered_batching_on_a_moe_checkpoint(tmp_path):
    print("week 3 batching, MoE checkpoint: max |log-prob difference|", case_week3_staggered_batching(tmp_path, moe=True))
