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
