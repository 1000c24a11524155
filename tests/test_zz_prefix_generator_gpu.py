"""GPU tier: ``KvPrefixGenerator`` (tiny_llm_hip/prefix.py; reference surface src/tiny_llm_ref/agent/branching.py:22-208) on the fused
decode engine: the checkpoint prefix is prefilled ONCE into a frozen slot, every continuation is tl_engine_fork (shared full pages,
copy-on-write tail) + suffix prefill + greedy decode through the captured graph.  A continuation must produce the ids the engine
produces for the whole steered prompt without any reuse, the prefill counter must show the prefix once, and the pool must be whole
again afterwards."""

import numpy as np
import pytest

from helpers import TINY_CFG, to_mlx_shaped
from oracle import tiny_oracle as O
from test_prefix_generator_cpu import MESSAGES, STEER_A, STEER_B, Tokenizer

pytestmark = pytest.mark.gpu


def test_forked_continuations_equal_generation_without_reuse():
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.prefix import KvPrefixGenerator, PrefixReuse

    w = O.make_qwen3_weights(TINY_CFG, seed=5, sigma=0.05)
    model = to_mlx_shaped(TINY_CFG, w)
    tok = Tokenizer()
    eng = DecodeEngine(model, page_size=16, num_pages=40, max_batch=2, max_prefill_rows=128)  # 16-token pages: the 44-token prefix ends inside a page
    try:
        gen = KvPrefixGenerator(eng, tok, max_tokens=8)
        cp = gen.save_checkpoint(MESSAGES)
        n = len(cp.cached_token_ids)
        assert n % 16 != 0, "the prefix must end inside a page so that the fork copies a tail page"
        base = eng.stats()
        for steer in (STEER_A, STEER_B, STEER_A):
            branch = gen.fork()
            branch.restore_checkpoint(cp)
            text = branch(MESSAGES + [steer])
            assert branch.reuse == PrefixReuse(n, cp.layer_offsets, n)
            full = tok.encode(tok.apply_chat_template(MESSAGES + [steer], add_generation_prompt=True))
            ids = eng.generate(full, 8, slot=0)  # no reuse: the whole prompt through the working slot
            want = []
            for t in ids:
                if t == tok.eos_token_id:
                    break
                want.append(t)
            assert text == tok.decode(want)
        st = eng.stats()
        # every continuation prefilled its suffix only (the no-reuse control runs prefilled whole prompts: counted apart)
        suffixes = sum(len(tok.encode(tok.apply_chat_template(MESSAGES + [s], add_generation_prompt=True))) - n for s in (STEER_A, STEER_B, STEER_A))
        controls = sum(len(tok.encode(tok.apply_chat_template(MESSAGES + [s], add_generation_prompt=True))) for s in (STEER_A, STEER_B, STEER_A))
        assert st["prefill_tokens"] - base["prefill_tokens"] == suffixes + controls
        assert eng.context_len(1) == n
        gen.close()
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()


def test_forked_continuation_ids_equal_the_checkers_on_a_peaked_checkpoint():
    """The comparand above is the same engine without reuse (self against self).  Here it is the CPU side: on a PEAKED checkpoint
    (tiny_llm_hip/synthetic.py: large embedding, damped residual writers, an untied head that is the embedding with its rows permuted --
    every step answers with a different id, with a top-2 margin far above the rounding error) a forked continuation must produce EXACTLY the
    greedy ids of the bf16 C port of the oracle (oracle/qwen3_decode.c) AND of the float64 truth (oracle/qwen3_truth.c) fed the whole
    steered prompt token by token.  Reference loop: src/tiny_llm_ref/agent/branching.py:133-156."""
    from oracle import c_oracle
    from helpers import log_parity, oracle_weights_from_model
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.prefix import KvPrefixGenerator
    from tiny_llm_hip.synthetic import synthetic_qwen3

    if not c_oracle.available():
        pytest.skip("oracle/libqwen3_oracle.so missing (run __graft_entry__.build())")
    model = synthetic_qwen3(TINY_CFG, seed=21, sigma=0.05, device="cuda", embed_sigma=0.5, residual_gain=0.5, head_permutation=(5, 11))
    weights = oracle_weights_from_model(model)
    cfg = dict(TINY_CFG, tie_word_embeddings=False)
    tok = Tokenizer()
    steps = 8
    eng = DecodeEngine(model, page_size=16, num_pages=40, max_batch=2, max_prefill_rows=128)
    try:
        gen = KvPrefixGenerator(eng, tok, max_tokens=steps)
        cp = gen.save_checkpoint(MESSAGES)
        for steer in (STEER_A, STEER_B):
            branch = gen.fork()
            branch.restore_checkpoint(cp)
            text = branch(MESSAGES + [steer])
            full = tok.encode(tok.apply_chat_template(MESSAGES + [steer], add_generation_prompt=True))
            want = {}
            for name, cls in (("bf16 port", c_oracle.COracleQwen3), ("float64 truth", c_oracle.CTruthQwen3)):
                m = cls(cfg, weights, max_ctx=len(full) + steps + 2)
                try:
                    tid, logits = 0, None
                    for t in full:
                        tid, logits = m.step(t)
                    ids, margins = [], []
                    for _ in range(steps):
                        if tid == tok.eos_token_id:
                            break
                        ids.append(int(tid))
                        top2 = np.partition(np.asarray(logits, dtype=np.float64), -2)[-2:]
                        margins.append(float(top2[1] - top2[0]))
                        tid, logits = m.step(tid)
                finally:
                    m.close()
                want[name] = (ids, min(margins) if margins else 0.0)
            log_parity({"what": "prefix_generator_peaked_ids", "engine": text, "port": want["bf16 port"][0], "truth": want["float64 truth"][0],
                        "min_top2_margin_truth": want["float64 truth"][1]})
            assert len(set(want["float64 truth"][0])) >= 4, f"the truth repeats itself: {want['float64 truth'][0]}"
            assert want["float64 truth"][1] > 0.5, f"the checkpoint does not discriminate: top-2 margin {want['float64 truth'][1]:.3f}"
            for name, (ids, _) in want.items():
                assert text == tok.decode(ids), f"forked continuation differs from the {name}: {text} vs {ids}"
        gen.close()
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()


def test_a_continuation_that_ends_before_the_room_does_is_not_an_error():
    """The decode loop runs in blocks of 16 steps; a block reserves room step by step.  A slot with room for fewer steps than the
    block asks for must still deliver a continuation whose end-of-sequence id arrives in time (the reference's loop goes token by
    token and never asks for more: agent/branching.py:145-156) -- and a continuation still going when the room ends must raise."""
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.prefix import KvPrefixGenerator

    w = O.make_qwen3_weights(TINY_CFG, seed=5, sigma=0.05)
    model = to_mlx_shaped(TINY_CFG, w)

    class EosTok(Tokenizer):
        eos_token_id = 2

    tok = EosTok()
    full = tok.encode(tok.apply_chat_template(MESSAGES + [STEER_A], add_generation_prompt=True))
    probe = DecodeEngine(model, page_size=16, num_pages=40, max_batch=2, max_prefill_rows=128)
    try:
        ids = probe.generate(full, 12, slot=0)
    finally:
        probe.close()
    # the 6th generated id becomes the end-of-sequence id; the working slot gets pages for the prompt + 8 tokens only
    tok.eos_token_id = int(ids[5])
    expected = []
    for t in ids:
        if t == tok.eos_token_id:
            break
        expected.append(t)
    page = 16
    n_prefix = len(tok.encode(tok.apply_chat_template(MESSAGES)))
    prefix_pages = (n_prefix + page - 1) // page
    total_pages = (len(full) + 8 + page - 1) // page
    own_pages = total_pages - n_prefix // page  # the fork shares the prefix's full pages and owns the rest (its copy of the tail page included)
    eng = DecodeEngine(model, page_size=page, num_pages=prefix_pages + own_pages, max_batch=2, max_prefill_rows=128)
    try:
        gen = KvPrefixGenerator(eng, tok, max_tokens=200)  # blocks of 16 steps do not fit: the slot's room ends 8..23 tokens behind the prompt
        cp = gen.save_checkpoint(MESSAGES)
        branch = gen.fork()
        branch.restore_checkpoint(cp)
        assert branch(MESSAGES + [STEER_A]) == tok.decode(expected)
        tok.eos_token_id = -1  # never produced: the continuation is still going when the room ends
        branch2 = gen.fork()
        branch2.restore_checkpoint(cp)
        with pytest.raises(RuntimeError):
            branch2(MESSAGES + [STEER_A])
        assert eng.context_len(1) == len(cp.cached_token_ids)  # the working slot was released, the frozen prefix is intact
        gen.close()
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()
