"""CPU tier: host-side logic above the extension — page pool, request caches, batching metadata, the paged_attention
wrapper's validation, chunked prefill and the continuous-batching scheduler with fault injection.

The extension itself is GPU-only; these tests route the two ops this logic touches through tests/fake_ext.py (oracle
backed, test-only).  Expected values are the reference's own known-answer literals
(tests/golden/reference_literals.json) and behaviours from tests_refsol/test_week_3_day_{1,2,3,4}.py.
"""

import json
from pathlib import Path

import numpy as np
import pytest
import torch

LIT = json.loads((Path(__file__).resolve().parent / "golden" / "reference_literals.json").read_text())
NEG = float("-inf")


def T(a, dtype=torch.float32):
    def conv(x):
        return [conv(v) for v in x] if isinstance(x, list) else (NEG if x == "-inf" else float(x))

    return torch.tensor(conv(a), dtype=dtype)


def chunk(n, heads=2, dim=4, seed=None):
    g = torch.Generator().manual_seed(n if seed is None else seed)
    return torch.randn((1, heads, n, dim), generator=g), torch.randn((1, heads, n, dim), generator=g)


# ------------------------------------------------------------------------------------------------ dense batching
def test_batching_kv_cache_dense_literals():
    from tiny_llm_hip import BatchingKvCache, TinyKvFullCache

    lit = LIT["batching_kv_cache"]
    cache = BatchingKvCache(max_active_requests=3)
    assert cache.max_seq_len is None
    slot0, slot2 = TinyKvFullCache(), TinyKvFullCache()
    slot0.update_and_fetch(T(lit["slot0"]["key"]), T(lit["slot0"]["value"]))
    slot2.update_and_fetch(T(lit["slot2"]["key"]), T(lit["slot2"]["value"]))
    cache.add_request(slot0, 0)
    cache.add_request(slot2, 2)
    keys, values, seq_len, mask = cache.update_and_fetch(T(lit["keys"]), T(lit["values"]), mask_length=2)
    assert seq_len is None
    assert torch.equal(keys, T(lit["expected_keys"]))
    assert torch.equal(values, T(lit["expected_values"]))
    assert torch.equal(mask, T(lit["expected_mask"]).reshape(3, 1, 2, 4))
    assert cache.last_batch_bytes == lit["last_batch_bytes"]
    assert cache.staging_copy_bytes == lit["staging_copy_bytes"]


# ------------------------------------------------------------------------------------------------ page pool
def test_paged_pool_growth_reset_and_reuse(cpu_ext):
    from tiny_llm_hip import TinyKvPagedCache, TinyKvPagedPool

    lit = LIT["paged_pool_growth"]
    pool = TinyKvPagedPool(page_size=lit["page_size"])
    cache = TinyKvPagedCache(pool=pool)
    cache.update_and_fetch_paged(*chunk(lit["tokens"]))
    assert pool.num_pages == lit["num_pages"] and pool.capacity == lit["capacity"]
    assert pool.key_pages.shape[0] == pool.num_pages == pool.value_pages.shape[0]
    assert pool.storage_growths == lit["storage_growths"]
    assert pool.copied_pages_on_growth == lit["copied_pages_on_growth"]
    assert pool.copied_bytes_on_growth == lit["copied_bytes_on_growth"]
    with pytest.raises(ValueError):
        pool.reset()  # a live request still owns pages (reference paged_kv_cache.py:184-186)
    cache.release()
    assert pool.capacity == 8 and pool.num_free_pages == 5 and pool.used_page_ids == set()
    # released pages are handed out again before the pool grows
    again = TinyKvPagedCache(pool=pool)
    again.update_and_fetch_paged(*chunk(6))
    assert pool.reused_page_allocations == 2 and pool.num_pages == 5
    again.release()
    pool.reset()
    assert (pool.capacity, pool.num_pages, pool.num_free_pages, pool.storage_nbytes) == (0, 0, 0, 0)
    assert (pool.storage_growths, pool.copied_pages_on_growth, pool.copied_bytes_on_growth) == (0, 0, 0)


def test_paged_cache_contents_block_table_identity_and_rewind(cpu_ext):
    from tiny_llm_hip import TinyKvPagedCache, TinyKvPagedPool

    pool = TinyKvPagedPool(page_size=4)
    blocker, cache = TinyKvPagedCache(pool=pool), TinyKvPagedCache(pool=pool)
    k1, v1 = chunk(3, seed=1)
    cache.update_and_fetch_paged(k1, v1)
    first = cache.block_table()
    assert cache.block_table() is first  # cached until the page ids change (reference paged_kv_cache.py:364-377)
    blocker.update_and_fetch_paged(*chunk(2, seed=2))  # takes physical page 1
    k2, v2 = chunk(1, seed=3)
    cache.update_and_fetch_paged(k2, v2)               # fills the tail of page 0
    assert cache.block_table() is first
    k3, v3 = chunk(6, seed=4)
    meta = cache.update_and_fetch_paged(k3, v3, mask="causal")  # needs two more pages -> ids [0, 2, 3]
    assert cache.block_table() is not first
    assert meta.block_table.tolist() == [[0, 2, 3]] and meta.context_lens.tolist() == [10]
    assert meta.block_table.dtype == torch.int32 and meta.context_lens.dtype == torch.int32
    dense_k, dense_v = cache.gather_dense()
    assert torch.equal(dense_k, torch.cat([k1, k2, k3], dim=2)) and torch.equal(dense_v, torch.cat([v1, v2, v3], dim=2))
    cache.rewind(5)  # drops page 3 and half of page 2 (reference paged_kv_cache.py:414-434)
    assert cache.offset == 5 and cache.block_table().tolist() == [[0, 2]]
    assert torch.equal(cache.gather_dense()[0], torch.cat([k1, k2, k3], dim=2)[:, :, :5])
    with pytest.raises(AssertionError):  # the reference asserts 0 <= n <= offset (paged_kv_cache.py:414-416)
        cache.rewind(6)
    cache.release()
    blocker.release()
    assert pool.used_page_ids == set()


def test_paged_append_is_transactional(cpu_ext, monkeypatch):
    """An injected write failure must leave pool and cache exactly as before (test_week_3_day_3.py:199-219)."""
    from tiny_llm_hip import TinyKvPagedCache, TinyKvPagedPool

    pool = TinyKvPagedPool(page_size=4)
    cache = TinyKvPagedCache(pool=pool)
    cache.update_and_fetch_paged(*chunk(3))
    before = (list(cache.page_ids), cache.offset, pool.num_pages, sorted(pool.used_page_ids), list(pool.free_page_ids))
    calls = {"n": 0}
    real = pool.write_page_slice

    def flaky(*args, **kwargs):
        calls["n"] += 1
        if calls["n"] == 2:
            raise RuntimeError("injected write failure")
        return real(*args, **kwargs)

    monkeypatch.setattr(pool, "write_page_slice", flaky)
    with pytest.raises(RuntimeError, match="injected"):
        cache.update_and_fetch_paged(*chunk(7))
    after = (list(cache.page_ids), cache.offset, pool.num_pages, sorted(pool.used_page_ids), list(pool.free_page_ids))
    assert after == before


# ------------------------------------------------------------------------------------------------ paged metadata
def test_paged_metadata_literals_and_attention_matches_dense(cpu_ext):
    from tiny_llm_hip import (BatchingKvCache, TinyKvPagedCache, TinyKvPagedPool, paged_attention,
                              scaled_dot_product_attention_grouped)

    single = LIT["paged_metadata_single"]
    pool = TinyKvPagedPool(page_size=single["page_size"])
    cache = TinyKvPagedCache(pool=pool)
    cache.update_and_fetch(*chunk(single["appends"][0], seed=1))
    key2, value2 = chunk(single["appends"][1], seed=2)
    meta = cache.update_and_fetch_paged(key2, value2, mask="causal")
    assert meta.block_table.tolist() == single["block_table"] and meta.context_lens.tolist() == single["context_lens"]
    assert tuple(meta.key_pages.shape) == (2, 2, 4, 4)
    q = torch.randn((1, 4, 3, 4), generator=torch.Generator().manual_seed(9))
    dense_k, dense_v = cache.gather_dense()
    want = scaled_dot_product_attention_grouped(q, dense_k, dense_v, mask="causal")
    got = paged_attention(q, meta.key_pages, meta.value_pages, meta.block_table, meta.context_lens, meta.page_size,
                          mask=meta.mask)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)

    batch_lit = LIT["paged_metadata_batch"]
    pool = TinyKvPagedPool(page_size=4)
    first, second = TinyKvPagedCache(pool=pool), TinyKvPagedCache(pool=pool)
    first.update_and_fetch(*chunk(batch_lit["prefilled"]["0"], seed=3))
    second.update_and_fetch(*chunk(batch_lit["prefilled"]["2"], seed=4))
    batch = BatchingKvCache(max_active_requests=3, max_seq_len=16)
    batch.add_request(first, 0)
    batch.add_request(second, 2)
    keys = torch.zeros((3, 2, 1, 4))
    values = torch.zeros((3, 2, 1, 4))
    keys[0:1], values[0:1] = chunk(1, seed=5)
    keys[2:3], values[2:3] = chunk(1, seed=6)
    meta = batch.update_and_fetch_paged(keys, values, mask_length=1, mask="causal")
    assert meta.context_lens.tolist() == batch_lit["context_lens"]
    assert tuple(meta.block_table.shape) == (3, 2) and meta.block_table.tolist()[1] == batch_lit["idle_row"]
    assert tuple(meta.key_pages.shape) == (3, 2, 4, 4)
    q = torch.randn((3, 4, 1, 4), generator=torch.Generator().manual_seed(10))
    out = paged_attention(q, meta.key_pages, meta.value_pages, meta.block_table, meta.context_lens, meta.page_size,
                          mask=meta.mask)
    for row, req in ((0, first), (2, second)):
        k, v = req.gather_dense()
        want = scaled_dot_product_attention_grouped(q[row:row + 1], k, v, mask="causal")
        torch.testing.assert_close(out[row:row + 1], want, rtol=1e-5, atol=1e-6)
    assert not out[1].any()  # idle slot


def test_batching_paged_update_rejects_mixed_pools_without_side_effects(cpu_ext):
    from tiny_llm_hip import BatchingKvCache, TinyKvPagedCache, TinyKvPagedPool

    first = TinyKvPagedCache(pool=TinyKvPagedPool(page_size=4))
    second = TinyKvPagedCache(pool=TinyKvPagedPool(page_size=4))
    first.update_and_fetch(*chunk(2, seed=1))
    second.update_and_fetch(*chunk(2, seed=2))
    batch = BatchingKvCache(max_active_requests=2, max_seq_len=8)
    batch.add_request(first, 0)
    batch.add_request(second, 1)
    state = lambda c: (list(c.page_ids), c.offset, sorted(c.pool.used_page_ids))  # noqa: E731
    before = (state(first), state(second))
    keys = torch.zeros((2, 2, 1, 4))
    with pytest.raises(ValueError, match="share one page pool"):
        batch.update_and_fetch_paged(keys, keys, mask_length=1)
    assert (state(first), state(second)) == before


@pytest.mark.parametrize("mutate,message", [
    (lambda m: m["ctx"].__setitem__(0, -1), "nonnegative"),
    (lambda m: m["ctx"].__setitem__(0, 9), "not covered"),
    (lambda m: m["table"].__setitem__((0, 1), 7), "outside physical page storage"),
    (lambda m: m["table"].__setitem__((0, 1), 0), "aliased"),
    (lambda m: (m["ctx"].__setitem__(0, 3), m["table"].__setitem__((0, 1), 1)), "must use the -1 sentinel"),
    (lambda m: m.__setitem__("page_size", 0), "positive integer"),
    (lambda m: (m["ctx"].__setitem__(0, 1), m["table"].__setitem__((0, 1), -1)), "must be zero or at least query length"),
])
def test_paged_attention_validation_messages(cpu_ext, mutate, message):
    """The wrapper's host-side checks and their message substrings (tests_refsol/test_week_3_day_4.py:248-322)."""
    from tiny_llm_hip import paged_attention

    m = {"ctx": torch.tensor([6], dtype=torch.int32), "table": torch.tensor([[0, 1]], dtype=torch.int32), "page_size": 4}
    mutate(m)
    q = torch.zeros((1, 2, 2, 4))
    pages = torch.zeros((2, 2, 4, 4))
    with pytest.raises(ValueError, match=message):
        paged_attention(q, pages, pages, m["table"], m["ctx"], m["page_size"], mask="causal")


def test_paged_attention_rejects_array_masks(cpu_ext):
    from tiny_llm_hip import paged_attention

    pages = torch.zeros((1, 1, 4, 4))
    with pytest.raises(NotImplementedError):
        paged_attention(torch.zeros((1, 1, 1, 4)), pages, pages, torch.zeros((1, 1), dtype=torch.int32),
                        torch.ones(1, dtype=torch.int32), 4, mask=torch.zeros((1, 1)))


# ------------------------------------------------------------------------------------------------ scheduler
class FakeDetokenizer:
    def __init__(self, _):
        self.text = ""

    def add_token(self, token):
        self.text += str(token)


class FakeTokenizer:
    eos_token_id = 99
    _tokenizer = object()
    detokenizer = FakeDetokenizer(_tokenizer)

    def encode(self, prompt, add_special_tokens=False):
        assert not add_special_tokens
        return list(range(1, len(prompt) + 1))


class FailingTextDetokenizer(FakeDetokenizer):
    def __init__(self, _):
        self._text = ""

    def add_token(self, token):
        self._text += str(token)

    @property
    def text(self):
        raise RuntimeError("injected detokenization failure")


class FailingTextTokenizer(FakeTokenizer):
    detokenizer = FailingTextDetokenizer(FakeTokenizer._tokenizer)


def make_paged_fake_model(output_token=1, fail_at=None):
    from tiny_llm_hip import BatchingKvCache, TinyKvPagedCache, TinyKvPagedPool

    class FailingMaterialize(TinyKvPagedCache):
        def materialize(self):
            super().materialize()
            raise RuntimeError("injected materialization failure")

    class Model:
        num_hidden_layers = 1

        def __init__(self):
            self.pool = TinyKvPagedPool(page_size=4)
            self.calls, self.cache_creations = [], 0

        def create_kv_cache(self):
            self.cache_creations += 1
            return [(FailingMaterialize if fail_at == "materialize" else TinyKvPagedCache)(self.pool)]

        def __call__(self, inputs, offsets, cache, logits_to_keep=1):
            offset = offsets[0] if isinstance(offsets, list) else int(offsets)
            n = len(self.calls) + 1
            self.calls.append((offset, inputs.shape[1]))
            key = torch.zeros((inputs.shape[0], 1, inputs.shape[1], 1))
            if isinstance(cache[0], BatchingKvCache):
                cache[0].update_and_fetch_paged(key, key, mask_length=inputs.shape[1])
            else:
                cache[0].update_and_fetch_paged(key, key)
            if fail_at == "prefill" and n == 1:
                raise RuntimeError("injected prefill failure")
            if fail_at == "decode" and n == 2:
                raise RuntimeError("injected decode failure")
            logits = torch.zeros((inputs.shape[0], 1, 128))
            logits[..., output_token] += 1
            return logits

    return Model()


def test_chunked_prefill_advances_in_bounded_steps(cpu_ext):
    from tiny_llm_hip import Request, TinyKvFullCache

    class Model:
        num_hidden_layers = 1
        calls = []

        def create_kv_cache(self):
            return [TinyKvFullCache()]

        def __call__(self, inputs, offsets, cache, logits_to_keep=1):
            self.calls.append((offsets[0], inputs.shape[1]))
            key = torch.zeros((1, 1, inputs.shape[1], 1))
            cache[0].update_and_fetch(key, key)
            logits = torch.zeros((1, 1, 4))
            logits[..., 1] += 1
            return logits

    model = Model()
    request = Request(model, FakeTokenizer(), "1234567", prefill_max_step=3, device="cpu")
    for want in (3, 6):
        request.try_prefill()
        assert request.offset == want == request.kv_cache[0].offset and not request.is_prefill_done
    request.try_prefill()
    assert request.offset == 7 and request.is_prefill_done and request.next_token == 1
    assert model.calls == [(0, 3), (3, 3), (6, 1)]
    with pytest.raises(ValueError, match="after done"):
        request.try_prefill()


def test_batch_generate_schedules_and_returns_every_page(cpu_ext):
    from tiny_llm_hip import batch_generate

    model = make_paged_fake_model()
    result = batch_generate(model, FakeTokenizer(), ["1234567"], max_seq_len=9, batch_size=1, prefill_step=3)
    assert result == [(0, "11")]
    assert model.calls == [(0, 3), (3, 3), (6, 1), (7, 1)]
    assert model.pool.used_page_ids == set() and model.pool.num_free_pages == model.pool.num_pages

    model = make_paged_fake_model(output_token=FakeTokenizer.eos_token_id)
    assert batch_generate(model, FakeTokenizer(), ["12345"], max_seq_len=10, batch_size=1, prefill_step=10) == [(0, "")]
    assert model.calls == [(0, 5)] and model.pool.used_page_ids == set()


@pytest.mark.parametrize("prompt,result,calls,creations", [("12", [(0, "1")], [(0, 2)], 1), ("123", [(0, "")], [(0, 3)], 1),
                                                            ("1234", None, [], 0)])
def test_batch_generate_enforces_max_seq_len(cpu_ext, prompt, result, calls, creations):
    from tiny_llm_hip import batch_generate

    model = make_paged_fake_model()
    if result is None:
        with pytest.raises(ValueError, match="exceeds max_seq_len"):
            batch_generate(model, FakeTokenizer(), [prompt], max_seq_len=3)
    else:
        assert batch_generate(model, FakeTokenizer(), [prompt], max_seq_len=3, batch_size=1) == result
    assert model.calls == calls and model.cache_creations == creations and model.pool.used_page_ids == set()


@pytest.mark.parametrize("failure_point", ["prefill", "materialize", "decode", "detokenize"])
def test_batch_generate_releases_pages_on_injected_failures(cpu_ext, failure_point):
    """Every live cache is released in the scheduler's finally block (reference batch.py:271-284)."""
    from tiny_llm_hip import batch_generate

    tokenizer = FailingTextTokenizer() if failure_point == "detokenize" else FakeTokenizer()
    model = make_paged_fake_model(fail_at=failure_point)
    with pytest.raises(RuntimeError, match="injected"):
        batch_generate(model, tokenizer, ["1"], max_seq_len=4, batch_size=1, prefill_step=4)
    assert model.pool.used_page_ids == set() and model.pool.num_free_pages == model.pool.num_pages


def test_batch_generate_interleaves_requests_across_slots(cpu_ext):
    from tiny_llm_hip import batch_generate

    model = make_paged_fake_model()
    result = batch_generate(model, FakeTokenizer(), ["12", "1234", "1"], max_seq_len=6, batch_size=2, prefill_step=2)
    assert sorted(idx for idx, _ in result) == [0, 1, 2]
    assert all(set(text) <= {"1"} and text for _, text in result)
    assert model.pool.used_page_ids == set() and model.pool.num_free_pages == model.pool.num_pages


class _FakeDecodeEngine:
    """Host-only stand-in for tiny_llm_hip.engine.DecodeEngine: every sequence emits prompt_sum + k for its k-th token, so
    the schedule of batch_generate_ids can be checked without a GPU."""

    def __init__(self, max_batch):
        self.max_batch = max_batch
        self.seq = {}          # slot -> {"base": int, "n": tokens produced so far}
        self.decode_rows = []  # rows argument of every decode call
        self.released = []

    def begin(self, slot):
        assert slot not in self.seq
        self.seq[slot] = {"base": 0, "n": 0}

    def prefill(self, slot, tokens, *, chunk, want_logits=True):
        assert len(tokens) <= chunk
        self.seq[slot]["base"] += sum(tokens)
        if want_logits:
            self.seq[slot]["n"] = 1

    def read_tokens(self, slot, n):
        s = self.seq[slot]
        return [s["base"] + k for k in range(s["n"] - n, s["n"])]

    def move(self, src, dst):
        assert dst not in self.seq
        self.seq[dst] = self.seq.pop(src)

    def decode(self, steps, batch):
        assert steps == 1
        self.decode_rows.append((batch, sorted(self.seq)))
        for slot, s in self.seq.items():
            if slot < batch:
                s["n"] += 1

    def read_pending(self, rows):
        return [self.seq[i]["base"] + self.seq[i]["n"] - 1 if i in self.seq else -1 for i in range(rows)]

    def release(self, slot):
        self.seq.pop(slot)
        self.released.append(slot)


def test_batch_generate_ids_decodes_only_the_occupied_slot_prefix():
    """engine.batch_generate_ids: same schedule as the reference (one prefill chunk + one decode step per turn), but a
    decode step covers a bucket over the occupied slot prefix instead of all batch_size rows."""
    from tiny_llm_hip.engine import _DECODE_ROW_BUCKETS, batch_generate_ids

    eng = _FakeDecodeEngine(max_batch=17)
    prompts = [[i + 1] * (3 + i % 5) for i in range(12)]
    limits = [2 + (i * 3) % 7 for i in range(12)]
    got = batch_generate_ids(eng, prompts, limits, batch_size=16, prefill_step=4)
    assert sorted(i for i, _ in got) == list(range(12))
    for idx, ids in got:
        base = sum(prompts[idx])
        assert ids == [base + k for k in range(limits[idx])], f"request {idx}"
    assert not eng.seq and sorted(set(eng.released)) == sorted(set(eng.released))
    assert eng.decode_rows, "no decode step ran"
    for rows, live in eng.decode_rows:
        occupied = [s for s in live if s < 16]
        assert rows in _DECODE_ROW_BUCKETS or rows == 16
        assert rows > max(occupied), "a live sequence was left out of the step"
        smaller = [b for b in _DECODE_ROW_BUCKETS if b < rows]
        assert not smaller or smaller[-1] <= max(occupied), "the step covered more rows than the bucket rule allows"
        # holes finished requests leave are closed whenever that lowers the bucket: a step runs at the bucket of the live COUNT
        assert rows == min(next(b for b in _DECODE_ROW_BUCKETS if b >= len(occupied)), 16), (rows, occupied)
    assert min(r for r, _ in eng.decode_rows) < 16, "short batches should not step all 16 rows"


# ---------------------------------------------------------------------------------------------------------------------
# speculative decoding (reference scenarios: tests_refsol/test_week_3_day_7.py) with position-scripted models
# ---------------------------------------------------------------------------------------------------------------------
class _SpecDetok:
    def __init__(self):
        self.tokens, self.finalized = [], 0

    def reset(self):
        self.tokens = []

    def add_token(self, t):
        self.tokens.append(int(t))

    def finalize(self):
        self.finalized += 1

    @property
    def text(self):
        return " ".join(str(t) for t in self.tokens)


class _SpecTokenizer:
    def __init__(self, eos=0, vocab=50, shift=0, with_vocab=True):
        self.eos_token_id = eos
        self._vocab = {str(i): i + shift for i in range(vocab)}
        self._detok = _SpecDetok()
        if not with_vocab:
            self.get_vocab = None

    def encode(self, prompt, add_special_tokens=False):
        return [int(x) for x in prompt.split()]

    def get_vocab(self):
        return dict(self._vocab)

    @property
    def detokenizer(self):
        return self._detok


class _SpecCache:
    def __init__(self):
        self.offset, self.released, self.rewinds = 0, 0, []

    def rewind(self, n):
        assert 0 < n <= self.offset
        self.offset -= n
        self.rewinds.append(n)

    def release(self):
        self.released += 1


class _ScriptedModel:
    """Predicts script[p + 1] at absolute position p, whatever it is fed (a wrong input only matters for rows the algorithm
    must discard).  Checks the cache offset on every call and logs (rows, logits_to_keep, dtype)."""

    def __init__(self, script, vocab=50, layers=2):
        self.script, self.vocab, self.layers = list(script), vocab, layers
        self.calls, self.caches = [], []

    def create_kv_cache(self):
        cache = [_SpecCache() for _ in range(self.layers)]
        self.caches.append(cache)
        return cache

    def __call__(self, tokens, offset, kv_cache, logits_to_keep=1):
        assert tokens.dtype == torch.int32 and tokens.dim() == 2 and tokens.shape[0] == 1
        rows = tokens.shape[1]
        for layer in kv_cache:
            assert layer.offset == offset, "model offset does not match the cache"
            layer.offset += rows
        self.calls.append((rows, logits_to_keep, offset))
        logits = torch.full((1, rows, self.vocab), -10.0)
        for i in range(rows):
            logits[0, i, self.script[offset + i + 1]] = 10.0
        return logits[:, -logits_to_keep:, :]


def _target_only_text(script, n_prompt, eos=0):
    out = []
    for t in script[n_prompt:]:
        if t == eos:
            break
        out.append(t)
    return " ".join(str(t) for t in out)


def _spec_run(target_script, draft_script, prompt="5 6 7", k=4):
    from tiny_llm_hip import speculative_generate

    target, draft = _ScriptedModel(target_script), _ScriptedModel(draft_script)
    tok, dtok = _SpecTokenizer(), _SpecTokenizer()
    text = speculative_generate(draft, target, dtok, tok, prompt, proposal_length=k, device="cpu")
    for m in (target, draft):
        for cache in m.caches:
            assert all(layer.released == 1 for layer in cache), "a KV cache was not released exactly once"
    return text, target, draft, tok


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("k", [1, 2, 4, 7])
def test_speculative_output_equals_target_only_decoding(seed, k):
    """Losslessness over random agreement patterns: whatever the draft proposes, the text is the target's own greedy text;
    every verification call feeds at most k + 1 rows and asks for all of their logits."""
    rng = np.random.default_rng(seed)
    n = 3
    length = int(rng.integers(4, 40))
    target = [5, 6, 7] + [int(t) for t in rng.integers(1, 50, size=length)] + [0] + [9] * 20
    draft = list(target)
    for p in range(n, len(draft)):
        if rng.random() < 0.35:
            draft[p] = int(rng.integers(0, 50))  # may also be an early EOS from the draft
    text, tm, dm, tok = _spec_run(target, draft, k=k)
    assert text == _target_only_text(target, n)
    assert tok.detokenizer.finalized == 1
    assert all(rows <= k + 1 and keep == rows for rows, keep, _ in tm.calls[1:])
    assert all(rows == 1 for rows, _, _ in dm.calls[1:])


def test_speculative_full_agreement_needs_few_target_calls():
    target = [5, 6, 7] + list(range(10, 34)) + [0] + [9] * 10
    text, tm, dm, _ = _spec_run(target, target, k=4)
    assert text == _target_only_text(target, 3)
    # 24 tokens + EOS with 5 rows per verification: prefill + ceil(25 / 5) calls, no rewinds on the target
    assert len(tm.calls) == 1 + 5
    assert all(not layer.rewinds for layer in tm.caches[0][:1]) or sum(tm.caches[0][0].rewinds) <= 5


def test_speculative_mismatch_rewinds_both_caches_to_the_accepted_prefix():
    target = [5, 6, 7, 11, 12, 13, 14, 15, 0] + [9] * 10
    draft = [5, 6, 7, 11, 12, 40, 41, 42, 43] + [9] * 10   # agrees on positions 3, 4 and diverges at 5
    text, tm, dm, _ = _spec_run(target, draft, k=4)
    assert text == "11 12 13 14 15"
    # first verification fed [11, 12, 40, 41, 42]: rows 0..1 accepted, row 2 is the first disagreement
    assert tm.calls[1] == (5, 5, 3)
    assert tm.caches[0][0].rewinds[0] == 3 and dm.caches[0][0].rewinds[0] == 2
    assert tm.calls[2][2] == 5  # the target resumes at absolute position 5 with its own token


def test_speculative_zero_proposals_never_builds_a_draft_cache_and_eos_first_stops_at_once():
    target = [5, 6, 7, 21, 22, 0] + [9] * 5
    text, tm, dm, _ = _spec_run(target, target, k=0)
    assert text == "21 22" and not dm.calls and not dm.caches
    assert all(rows == 1 for rows, _, _ in tm.calls[1:])
    text, tm, dm, _ = _spec_run([5, 6, 7, 0] + [9] * 5, [5, 6, 7, 30] + [9] * 5, k=3)
    assert text == "" and len(tm.calls) == 1 and not dm.calls


def test_speculative_draft_eos_at_prefill_falls_back_to_target_only():
    target = [5, 6, 7, 21, 22, 23, 0] + [9] * 5
    draft = [5, 6, 7, 0] + [9] * 8
    text, tm, dm, _ = _spec_run(target, draft, k=3)
    assert text == "21 22 23" and len(dm.calls) == 1
    assert all(rows == 1 for rows, _, _ in tm.calls[1:])


@pytest.mark.parametrize("case,message", [
    ("encode", "encode the prompt differently"), ("eos", "different EOS token ids"),
    ("novocab", "comparable vocabularies"), ("vocab", "different token ids"), ("empty", "at least one token"),
    ("k", "non-negative integer"), ("kbool", "non-negative integer")])
def test_speculative_validation_runs_before_any_model_call(case, message):
    from tiny_llm_hip import speculative_generate

    target, draft = _ScriptedModel([1] * 20), _ScriptedModel([1] * 20)
    tok, dtok, prompt, k = _SpecTokenizer(), _SpecTokenizer(), "5 6 7", 2
    if case == "encode":
        dtok.encode = lambda p, add_special_tokens=False: [5, 6]
    elif case == "eos":
        dtok.eos_token_id = 3
    elif case == "novocab":
        dtok = _SpecTokenizer(with_vocab=False)
    elif case == "vocab":
        dtok = _SpecTokenizer(shift=1)
    elif case == "empty":
        prompt = ""
    elif case == "k":
        k = -1
    elif case == "kbool":
        k = True
    with pytest.raises(ValueError, match=message):
        speculative_generate(draft, target, dtok, tok, prompt, proposal_length=k, device="cpu")
    assert not target.calls and not draft.calls and not target.caches
