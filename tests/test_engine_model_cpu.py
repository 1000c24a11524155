"""CPU tier: the adapter that puts the fused decode engine behind the course model's call surface
(tiny_llm_hip/engine_model.py), driven exactly as the reference's single-request bench loop drives a Week-2 model
(benches/bench.py:run_one_request_week2, 277-312: create_kv_cache -> model(prompt[None], 0, cache, logits_to_keep) ->
argmax of [:, -1, :] -> model(token[None], offset, cache) ... -> release every layer's cache).  The engine itself needs a GPU;
here an engine with the same Python API answers from the numpy oracle, so what is tested is the adapter: call sequence,
offsets, slot hand-back, refusals.  On the GPU the same adapter runs on DecodeEngine (tests/test_zz_engine_model_gpu.py)."""

import numpy as np
import pytest
import torch

from helpers import TINY_CFG, to_mlx_shaped
from oracle import tiny_oracle as O


class OracleEngine:
    """DecodeEngine's request API (begin / prefill / set_token / decode / logits / context_len / rewind / release) on
    OracleQwen3.  Records the calls it receives."""

    def __init__(self, mlx_model, *, page_size, num_pages, max_batch, max_pages_per_seq, max_prefill_rows):
        self.kwargs = dict(page_size=page_size, num_pages=num_pages, max_batch=max_batch, max_pages_per_seq=max_pages_per_seq,
                           max_prefill_rows=max_prefill_rows)
        self.weights = mlx_model.oracle_weights
        self.model, self.pending, self.last, self.calls = None, None, None, []

    def begin(self, slot=0):
        assert slot == 0 and self.model is None
        self.model = O.OracleQwen3(TINY_CFG, self.weights)
        self.calls.append("begin")

    def context_len(self, slot=0):
        return self.model.offset if self.model is not None else 0

    def prefill(self, slot, tokens, *, chunk=2048, want_logits=True):
        self.last = self.model.forward(list(tokens))[0, -1]
        self.pending = int(np.argmax(self.last))
        self.calls.append(("prefill", len(tokens)))

    def set_token(self, slot, token):
        self.pending = int(token)

    def decode(self, steps, batch=None, use_graph=True):
        assert steps == 1 and batch == 1
        self.last = self.model.forward([self.pending])[0, -1]
        self.pending = int(np.argmax(self.last))
        self.calls.append("decode")

    def logits(self, rows=1):
        return torch.from_numpy(np.asarray(self.last, dtype=np.float32)[None]).to(torch.bfloat16)

    def release(self, slot=0):
        assert self.model is not None
        self.model = None
        self.calls.append("release")

    def close(self):
        self.calls.append("close")


@pytest.fixture()
def fused():
    from tiny_llm_hip.engine_model import Qwen3ModelFused

    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    mlx_model = to_mlx_shaped(TINY_CFG, w, device="cpu")
    mlx_model.oracle_weights = w
    return Qwen3ModelFused(mlx_model, page_size=16, max_context=200, engine_factory=OracleEngine), w


def harness_request(model, prompt, max_new_tokens, prefill_logits_to_keep=None):
    """The reference's loop, call for call (benches/bench.py:277-312), on torch tensors."""
    cache = model.create_kv_cache()
    try:
        context = torch.tensor(prompt, dtype=torch.int32)
        logits = model(context[None, :], 0, cache, logits_to_keep=prefill_logits_to_keep)
        token = torch.argmax(logits[:, -1, :], dim=-1)
        out, offset = [int(token)], len(prompt)
        for _ in range(max_new_tokens - 1):
            logits = model(token.to(torch.int32)[None, :], offset, cache, logits_to_keep=1)
            token = torch.argmax(logits[:, -1, :], dim=-1)
            out.append(int(token))
            offset += 1
        return out
    finally:
        for layer_cache in cache:
            layer_cache.release()


def test_harness_loop_on_the_engine_backed_model(fused):
    model, w = fused
    assert model.engine.kwargs == dict(page_size=16, num_pages=14, max_batch=1, max_pages_per_seq=14, max_prefill_rows=2048)
    prompt = [int(t) for t in np.random.default_rng(5).integers(1, TINY_CFG["vocab_size"], size=23)]
    got = harness_request(model, prompt, 7)
    ref = O.OracleQwen3(TINY_CFG, w)
    want = [int(np.argmax(ref.forward(prompt)[0, -1]))]
    for _ in range(6):
        want.append(int(np.argmax(ref.forward([want[-1]])[0, -1])))
    assert got == want
    assert model.engine.calls == ["begin", ("prefill", 23)] + ["decode"] * 6 + ["release"]  # ONE release for the 2 layer handles
    # the slot is free again: a second request runs, and sees its own context
    assert harness_request(model, prompt[:5], 3, prefill_logits_to_keep=1) == harness_request(model, prompt[:5], 3)
    assert model.engine.calls.count("release") == 3


def test_engine_backed_model_refuses_what_the_engine_cannot_do(fused):
    model, _ = fused
    cache = model.create_kv_cache()
    assert len(cache) == TINY_CFG["num_hidden_layers"] and cache[0].slot == 0 and cache[1].offset == 0
    with pytest.raises(RuntimeError, match="one request at a time"):
        model.create_kv_cache()
    with pytest.raises(ValueError, match="one request per call"):
        model(torch.zeros((2, 1), dtype=torch.int32), 0, cache)
    model(torch.tensor([[5, 6, 7]], dtype=torch.int32), 0, cache)
    assert cache[1].offset == 3
    with pytest.raises(ValueError, match="in order"):
        model(torch.tensor([[9]], dtype=torch.int32), 7, cache)
    logits = model(torch.tensor([[9]], dtype=torch.int32), torch.tensor([3]), cache)  # offsets may arrive as arrays
    assert tuple(logits.shape) == (1, 1, TINY_CFG["vocab_size"]) and logits.dtype == torch.bfloat16
    for layer_cache in cache:
        layer_cache.release()
    with pytest.raises(ValueError, match="not be released"):
        model(torch.tensor([[1]], dtype=torch.int32), 0, cache)
    model.close()
    assert model.engine.calls[-1] == "close"


def test_dispatch_hands_out_the_engine_backed_model_on_request(monkeypatch):
    import tiny_llm_hip.engine_model as em
    import tiny_llm_hip.models as models

    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    mlx_model = to_mlx_shaped(TINY_CFG, w, device="cpu")
    mlx_model.oracle_weights = w
    made = []

    class Probe(em.Qwen3ModelFused):
        def __init__(self, mlx_model, **kwargs):
            made.append(kwargs)
            super().__init__(mlx_model, engine_factory=OracleEngine, **kwargs)

    monkeypatch.setattr(em, "Qwen3ModelFused", Probe)
    monkeypatch.setenv("TINY_LLM_FUSED_ENGINE", "1")
    assert isinstance(models.dispatch_model("qwen3-4b", mlx_model, week=3, enable_paged_attention=True), Probe)
    assert isinstance(models.dispatch_model("qwen3-4b", mlx_model, week=2), Probe)
    assert made == [dict(enable_paged_attention=True), dict()]
    # a named Week-2 checkpoint is a request for THAT op-by-op model; Week 1 has no KV cache at all
    assert type(models.dispatch_model("qwen3-4b", mlx_model, week=1)).__name__ == "Qwen3ModelWeek1"
    monkeypatch.setenv("TINY_LLM_FUSED_ENGINE", "0")
    assert not em.fused_engine_requested()
