"""GPU tier: Qwen3-MoE layers inside the fused engine (tl_engine_set_moe_layer; reference: the Week-3 model builds a Moe block for
every sparse layer, src/tiny_llm_ref/qwen3_week3.py:209-214, 258-272; the block is moe.py:39-89).

A 3-layer checkpoint whose layers 1 and 2 are sparse (4 experts, top 2, renormalised scores; layer 0 dense through mlp_only_layers)
is written to disk in the mlx_lm layout, loaded by tiny_llm_hip.load, and run two ways on the same weights: op by op through the
host mirror (Qwen3ModelWeek3 with tiny_llm_hip.Moe blocks -- held against the numpy oracle's moe_block by tests/test_ops_gpu.py and
against the reference's own classes through the facade by tests/test_loader_cpu.py) and through the fused engine, whose MoE layers
run the same op sequence inside the captured step.  Teacher-forced, so both sides see the same history: the logits must agree
within a few bf16 steps at every position, prefill by GEMV rows (<= 8), by GEMM rows, and chunked; decode by graph replay."""

import numpy as np
import pytest
import torch

from helpers import TINY_CFG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def moe_model(tmp_path_factory):
    from checkpoint_fixture import MOE_CFG_OVERRIDES, make_moe_weights, write_checkpoint
    from tiny_llm_hip import load

    cfg = dict(TINY_CFG, **MOE_CFG_OVERRIDES)
    w = make_moe_weights(cfg, seed=21)
    words = [f"w{i}" for i in range(cfg["vocab_size"] - 2)]
    path = write_checkpoint(tmp_path_factory.mktemp("moe") / "ckpt", cfg, w, vocab_words=words)
    model, _ = load(str(path))
    model._oracle_weights = w  # the numpy weight dict the checkpoint was written from (oracle / truth side of the tests below)
    return cfg, model


def mirror_logits(model, prompt, forced):
    from tiny_llm_hip import Moe, Qwen3ModelWeek3

    week3 = Qwen3ModelWeek3(model, page_size=16)
    assert [type(layer.mlp).__name__ for layer in week3.layers_inner] == ["Qwen3MLP", "Moe", "Moe"]
    assert isinstance(week3.layers_inner[1].mlp, Moe)
    cache = week3.create_kv_cache()
    try:
        toks = torch.tensor([prompt], dtype=torch.int32, device="cuda")
        out = [week3(toks, 0, cache, logits_to_keep=1)[0, -1].float().cpu().numpy()]
        offset = len(prompt)
        for tok in forced:
            t = torch.tensor([[tok]], dtype=torch.int32, device="cuda")
            out.append(week3(t, offset, cache, logits_to_keep=1)[0, -1].float().cpu().numpy())
            offset += 1
        return np.stack(out)
    finally:
        for c in cache:
            c.release()


def engine_logits(model, prompt, forced, chunk, batch_slot=0):
    from tiny_llm_hip.engine import DecodeEngine

    eng = DecodeEngine(model, page_size=16, num_pages=32, max_batch=2, max_prefill_rows=64)
    try:
        eng.begin(batch_slot)
        eng.prefill(batch_slot, prompt, chunk=chunk)
        got = [eng.logits(1)[0].float().cpu().numpy()]
        for tok in forced:
            eng.set_token(batch_slot, tok)
            eng.decode(1, batch=1)
            got.append(eng.logits(1)[0].float().cpu().numpy())
        stats = eng.stats()
        eng.release(batch_slot)
        return np.stack(got), stats
    finally:
        eng.close()


@pytest.mark.parametrize("n_prompt,chunk", [(5, 64), (23, 64), (50, 16)])
def test_fused_engine_with_moe_layers_matches_the_op_by_op_model(moe_model, n_prompt, chunk):
    cfg, model = moe_model
    rng = np.random.default_rng(n_prompt)
    prompt = [int(t) for t in rng.integers(2, cfg["vocab_size"], size=n_prompt)]
    forced = [int(t) for t in rng.integers(2, cfg["vocab_size"], size=6)]
    want = mirror_logits(model, prompt, forced)
    got, stats = engine_logits(model, prompt, forced, chunk)
    assert stats["graph_replays"] + stats["graph_captures"] >= len(forced), "the decode steps did not go through captured graphs"
    assert np.isfinite(got).all()
    step = 2.0 ** -7 * max(1.0, float(np.abs(want).max()))  # one bf16 step of the largest logit
    worst = float(np.abs(got - want).max())
    # the two paths share every expert kernel; they differ where the dense paths differ (fused GEMV against op-by-op launches,
    # the fused attention) and in the softmax / accumulation order of the routing arithmetic
    assert worst <= 4 * step, f"logits {worst / step:.1f} bf16 steps apart (prompt {n_prompt}, chunk {chunk})"
    assert (np.argmax(got, -1) == np.argmax(want, -1)).mean() >= 0.8, "greedy ids mostly differ"


def test_moe_layers_must_be_attached_before_the_first_step(moe_model):
    """tl_engine_set_moe_layer after a prefill is refused (captured graphs and workspaces are fixed by then)."""
    from tiny_llm_hip.engine import DecodeEngine

    cfg, model = moe_model
    eng = DecodeEngine(model, page_size=16, num_pages=8, max_batch=1, max_prefill_rows=16)
    try:
        eng.begin(0)
        eng.prefill(0, [5, 6, 7])
        with pytest.raises(RuntimeError, match="before the first prefill"):
            eng._attach_moe(1, model.model.layers[1].mlp, model.args)
    finally:
        eng.close()


def test_two_sequences_decode_together_as_they_do_alone(moe_model):
    """Batched decode over MoE layers: rows x top_k expert rows in one grouped launch; each sequence's logits stay within two bf16
    steps of its own single-sequence run (the dense projections take the two-row GEMV plan) and free-running greedy ids repeat."""
    from tiny_llm_hip.engine import DecodeEngine

    cfg, model = moe_model
    rng = np.random.default_rng(5)
    prompts = [[int(t) for t in rng.integers(2, cfg["vocab_size"], size=n)] for n in (9, 30)]
    forced = [[int(t) for t in rng.integers(2, cfg["vocab_size"], size=5)] for _ in prompts]
    alone = [engine_logits(model, p, f, 64)[0] for p, f in zip(prompts, forced)]
    eng = DecodeEngine(model, page_size=16, num_pages=32, max_batch=2, max_prefill_rows=64)
    try:
        for slot, p in enumerate(prompts):
            eng.begin(slot)
            eng.prefill(slot, p, chunk=64)
        got = [[], []]
        for s in range(5):
            for slot in range(2):
                eng.set_token(slot, forced[slot][s])
            eng.decode(1, batch=2)
            lg = eng.logits(2).float().cpu().numpy()
            for slot in range(2):
                got[slot].append(lg[slot])
        for slot in range(2):
            eng.release(slot)
    finally:
        eng.close()
    for slot in range(2):
        want = alone[slot][1:]
        have = np.stack(got[slot])
        step = 2.0 ** -7 * max(1.0, float(np.abs(want).max()))
        assert float(np.abs(have - want).max()) <= 2 * step, f"sequence {slot}: batched and single-sequence logits differ"


@pytest.mark.parametrize("n_prompt,chunk", [(5, 64), (7, 64), (23, 64)])
def test_fused_engine_with_moe_layers_against_the_oracle_and_the_float64_truth(moe_model, n_prompt, chunk):
    """The comparand of the first test is the HIP op-by-op model (the same expert kernels on both sides).  Here it is the CPU side: the
    numpy oracle with Qwen3-MoE layers (oracle.OracleQwen3 + moe_block: src/tiny_llm_ref/moe.py:39-89 inside qwen3_week3.py:55-121)
    and the float64 truth (oracle.TruthQwen3, pinned to transformers' Qwen3MoeForCausalLM by tests/test_truth_vs_transformers_cpu.py),
    teacher-forced on the same tokens.  The top-k choice is the model's one discontinuity: the truth is evaluated on the ORACLE's
    choice, and a position counts only where the truth's own choice is the same set of experts (a toss-up between the k-th and the
    (k+1)-th expert is not an error of either side); at least two thirds of the positions must count.  Bar: the engine sits as close to
    the truth as the bf16 oracle does (helpers.check_against_truth); 5 and 7 prompt tokens prefill by GEMV rows, 23 by the tile GEMM."""
    from helpers import check_against_truth
    from oracle import tiny_oracle as O

    cfg, model = moe_model
    w = model._oracle_weights
    rng = np.random.default_rng(100 + n_prompt)
    prompt = [int(t) for t in rng.integers(2, cfg["vocab_size"], size=n_prompt)]
    forced = [int(t) for t in rng.integers(2, cfg["vocab_size"], size=6)]
    got, stats = engine_logits(model, prompt, forced, chunk)
    assert stats["graph_replays"] + stats["graph_captures"] >= len(forced)
    oracle, truth = O.OracleQwen3(cfg, w), O.TruthQwen3(cfg, w)
    want = [oracle.forward(prompt)[0, -1]] + [oracle.forward([t])[0, -1] for t in forced]
    truth.forced_moe_ids = oracle.moe_ids
    exact = [truth.forward(prompt)[0, -1]] + [truth.forward([t])[0, -1] for t in forced]
    # call c of a sparse layer = position c of the teacher-forced sequence (call 0: the prompt, its last token is the row compared)
    counted = [all(bool(calls[c]["same_choice"][-1]) for calls in truth.moe_margins.values()) for c in range(len(forced) + 1)]
    assert sum(counted) >= 2 * len(counted) // 3 + 1, f"too many toss-up selections to compare: {counted}"
    keep = np.asarray(counted)
    check_against_truth(np.asarray(got)[keep], np.stack(want)[keep], np.stack(exact)[keep],
                        what=f"fused engine with 2 MoE layers, prompt {n_prompt} (chunk {chunk}), {int(keep.sum())} of {len(keep)} positions")
