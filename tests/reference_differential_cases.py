"""DIFFERENTIAL cases: the reference's OWN Python sources (/root/reference/src/tiny_llm_ref: operators, caches, page pool,
models, scheduler, speculative decoding) and the product's host mirror (tiny_llm_hip), imported side by side in one process
and driven with identical inputs over identical backends:

    * `mlx.core` / `mlx.nn` / `mlx_lm` -> the torch facade (tiny-llm_amd/compat) for BOTH,
    * `extensions_ref.tiny_llm_ext_ref` (the reference's Metal extension) -> the product's C-ABI binding for the reference too,
      answered in this container by the numpy oracle (tests/refsol_oracle_plugin.py).

Whatever the two code bases compute, they compute it from the same primitive results -- so every difference is a difference in
the HOST LOGIC (wiring, dtypes, rounding points, masks, offsets, cache and page bookkeeping, scheduling, acceptance rules).
The assertion is BIT EQUALITY of every tensor and equality of every counter / id / text.

Not collected by `pytest tests/` (file name); started by tests/test_reference_differential_cpu.py as
`pytest tests/reference_differential_cases.py -p refsol_oracle_plugin`.  Needs /root/reference (build container only).
"""

import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path("/root/reference")
if not (REFERENCE / "src" / "tiny_llm_ref").is_dir():
    pytest.skip("/root/reference is not present", allow_module_level=True)

import tiny_llm_ext_hip  # noqa: E402  (patched by refsol_oracle_plugin: the oracle answers the C ABI)

_pkg = types.ModuleType("extensions_ref")
_pkg.__path__ = []
_pkg.tiny_llm_ext_ref = tiny_llm_ext_hip
sys.modules["extensions_ref"] = _pkg
sys.modules["extensions_ref.tiny_llm_ext_ref"] = tiny_llm_ext_hip
for _name in [k for k in sys.modules if k == "tiny_llm_ref" or k.startswith("tiny_llm_ref.")]:
    del sys.modules[_name]  # the facade's alias package must not stand in for the real sources here
sys.path.insert(0, str(REFERENCE / "src"))
sys.path.insert(1, str(REFERENCE))

import mlx.core as mx  # noqa: E402
import tiny_llm_hip as P  # noqa: E402  the product
import tiny_llm_ref as R  # noqa: E402  the reference's sources

from checkpoint_fixture import MOE_CFG_OVERRIDES, make_moe_weights, write_stand_in_checkpoints  # noqa: E402
from helpers import TINY_CFG, to_mlx_shaped  # noqa: E402
from oracle import tiny_oracle as O  # noqa: E402

assert Path(R.__file__).is_relative_to(REFERENCE) and Path(P.__file__).is_relative_to(ROOT)


def same(a, b, what=""):
    """Bit equality, recursively over tuples / lists / dicts; tensors must agree in dtype and shape too."""
    def fail(msg):
        raise AssertionError(f"{what}: {msg}")

    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
            fail(f"{type(a).__name__} vs {type(b).__name__}")
        if a.dtype != b.dtype or tuple(a.shape) != tuple(b.shape):
            fail(f"{a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}")
        if not torch.equal(a, b):
            af, bf = a.double(), b.double()
            if not (torch.equal(torch.isnan(af), torch.isnan(bf)) and torch.equal(torch.nan_to_num(af, posinf=9e99, neginf=-9e99),
                                                                                 torch.nan_to_num(bf, posinf=9e99, neginf=-9e99))):
                fail(f"{int((af != bf).sum())} of {a.numel()} elements differ, max |difference| {float(torch.nan_to_num(af - bf).abs().max()):.3e}")
    elif isinstance(a, (tuple, list)):
        if type(a) is not type(b) or len(a) != len(b):
            fail(f"{type(a).__name__}[{len(a)}] vs {type(b).__name__}[{len(b) if hasattr(b, '__len__') else '?'}]")
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f"{what}[{i}]")
    elif isinstance(a, dict):
        if a.keys() != b.keys():
            fail(f"keys {sorted(a)} vs {sorted(b)}")
        for k in a:
            same(a[k], b[k], f"{what}[{k!r}]")
    elif a != b:
        fail(f"{str(a)[:200]!r} vs {str(b)[:200]!r}")


@pytest.fixture(scope="module")
def checkpoint():
    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    with mx.stream(mx.cpu):
        return to_mlx_shaped(TINY_CFG, w, device="cpu")


def prompt_ids(n, seed):
    return [int(t) for t in np.random.default_rng(seed).integers(1, TINY_CFG["vocab_size"], size=n)]


# ---- operators ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
def test_readable_operators(dtype):
    with mx.stream(mx.cpu):
        mx.random.seed(1)
        x = mx.random.normal((2, 5, 64)).astype(dtype)
        w = mx.random.normal((48, 64)).astype(dtype)
        b = mx.random.normal((48,)).astype(dtype)
        same(R.softmax(x, axis=-1), P.softmax(x, axis=-1), "softmax")
        same(R.linear(x, w), P.linear(x, w), "linear")
        same(R.linear(x, w, b), P.linear(x, w, b), "linear + bias")
        same(R.silu(x), P.silu(x), "silu")
        for L, S in ((4, 4), (3, 7), (1, 9)):
            same(R.causal_mask(L, S, dtype), P.causal_mask(L, S, dtype), f"causal_mask({L}, {S})")
        q = mx.random.normal((2, 8, 5, 32)).astype(dtype)
        k = mx.random.normal((2, 2, 9, 32)).astype(dtype)
        v = mx.random.normal((2, 2, 9, 32)).astype(dtype)
        array_mask = mx.random.normal((2, 8, 5, 9)).astype(dtype)
        for mask in (None, "causal", array_mask):
            for scale in (None, 0.2):
                same(R.scaled_dot_product_attention_grouped(q, k, v, scale, mask), P.scaled_dot_product_attention_grouped(q, k, v, scale, mask),
                     f"grouped attention, mask {type(mask).__name__}, scale {scale}")
        same(R.scaled_dot_product_attention_simple(k, k, v), P.scaled_dot_product_attention_simple(k, k, v), "simple attention")
        weight = (1 + 0.1 * mx.random.normal((64,))).astype(dtype)
        same(R.RMSNorm(64, weight, eps=1e-5)(x), P.RMSNorm(64, weight, eps=1e-5)(x), "RMSNorm")
        h = mx.random.normal((2, 6, 4, 32)).astype(dtype)  # [B, L, H, D]
        for traditional in (False, True):
            ref, own = R.RoPE(32, 64, 10000, traditional=traditional), P.RoPE(32, 64, 10000, traditional=traditional)
            same(ref(h), own(h), f"RoPE traditional={traditional}")
            same(ref(h, offset=slice(5, 11)), own(h, offset=slice(5, 11)), f"RoPE traditional={traditional}, offset slice")
        table = mx.random.normal((50, 64)).astype(dtype)
        ids = mx.array([[1, 49, 7], [0, 3, 3]], dtype=mx.int32)
        same(R.Embedding(50, 64, table)(ids), P.Embedding(50, 64, table)(ids), "Embedding")
        same(R.Embedding(50, 64, table).as_linear(x), P.Embedding(50, 64, table).as_linear(x), "Embedding.as_linear")
        heads = 4
        wq, wk, wv, wo = (mx.random.normal((64, 64)).astype(dtype) * 0.1 for _ in range(4))
        same(R.SimpleMultiHeadAttention(64, heads, wq, wk, wv, wo)(x, x, x, mask=None),
             P.SimpleMultiHeadAttention(64, heads, wq, wk, wv, wo)(x, x, x, mask=None), "SimpleMultiHeadAttention")


def test_quantized_operators_and_kernel_wrappers(checkpoint):
    with mx.stream(mx.cpu):
        mx.random.seed(2)
        layer = checkpoint.model.layers[0].self_attn.q_proj
        same(R.dequantize_linear(layer), P.dequantize_linear(layer), "dequantize_linear")
        x1 = mx.random.normal((1, 256)).astype(mx.bfloat16)
        x40 = mx.random.normal((40, 256)).astype(mx.bfloat16)
        for flags in (dict(), dict(use_simdgroup_matmul=True), dict(use_simdgroup_matmul=True, use_split_k_matmul=True),
                      dict(use_simdgroup_matvec=False)):
            rw, pw = R.QuantizedWeights.from_mlx_layer(layer, **flags), P.QuantizedWeights.from_mlx_layer(layer, **flags)
            for x in (x1, x40, x40.reshape(2, 20, 256)):
                same(R.quantized_linear(x, rw), P.quantized_linear(x, pw), f"quantized_linear {flags} {tuple(x.shape)}")
        rw, pw = R.QuantizedWeights.from_mlx_layer(layer), P.QuantizedWeights.from_mlx_layer(layer)
        args = (rw.scales, rw.biases, 128, 4, x1, rw.weight)
        same(R.quantized_matmul_vanilla(*args, True), P.quantized_matmul_vanilla(*args, True), "quantized_matmul_vanilla")
        same(R.quantized_matvec_custom(*args, True), P.quantized_matvec_custom(*args, True), "quantized_matvec_custom")
        emb = checkpoint.model.embed_tokens
        ids = mx.array([[5, 1000, 17]], dtype=mx.int32)
        for custom in (False, True):
            re = R.QuantizedEmbedding(1024, 256, R.QuantizedWeights.from_mlx_layer(emb), use_custom_kernel=custom)
            pe = P.QuantizedEmbedding(1024, 256, P.QuantizedWeights.from_mlx_layer(emb), use_custom_kernel=custom)
            same(re(ids), pe(ids), f"QuantizedEmbedding custom={custom}")
            same(re.as_linear(x1), pe.as_linear(x1), f"QuantizedEmbedding.as_linear custom={custom}")
        w = (1 + 0.1 * mx.random.normal((256,))).astype(mx.bfloat16)
        same(R.FastRMSNorm(256, w, 1e-6)(x40), P.FastRMSNorm(256, w, 1e-6)(x40), "FastRMSNorm")
        h = mx.random.normal((3, 2, 4, 128)).astype(mx.bfloat16)
        for off in (0, [3, 0, 9], mx.array([1, 2, 3], dtype=mx.int32)):
            same(R.FastRoPE(128, 64, 1000000)(h, off), P.FastRoPE(128, 64, 1000000)(h, off), f"FastRoPE offset {off!r}")
        g, u = mx.random.normal((7, 96)).astype(mx.bfloat16), mx.random.normal((7, 96)).astype(mx.bfloat16)
        same(R.swiglu(g, u), P.swiglu(g, u), "swiglu")
        q = mx.random.normal((2, 4, 1, 128)).astype(mx.bfloat16)
        k = mx.random.normal((2, 2, 33, 128)).astype(mx.bfloat16)
        v = mx.random.normal((2, 2, 33, 128)).astype(mx.bfloat16)
        for mask in (None, "causal"):
            same(R.decode_attention_custom(q, k, v, scale=0.1, mask=mask), P.decode_attention_custom(q, k, v, scale=0.1, mask=mask),
                 f"decode_attention_custom mask={mask}")


# ---- caches and the page pool --------------------------------------------------------------------------------------------
def cache_state(c):
    keep = {}
    for name, value in vars(c).items():
        if isinstance(value, (int, float, bool, str, type(None), torch.Tensor, list, tuple, dict)):
            keep[name] = value
    return keep


def test_dense_and_batching_caches():
    with mx.stream(mx.cpu):
        mx.random.seed(3)
        chunks = [(mx.random.normal((1, 2, n, 16)).astype(mx.bfloat16), mx.random.normal((1, 2, n, 16)).astype(mx.bfloat16)) for n in (5, 1, 1, 3)]
        rc, pc = R.TinyKvFullCache(), P.TinyKvFullCache()
        for i, (k, v) in enumerate(chunks):
            for mask in (None, "causal"):
                pass
            same(rc.update_and_fetch(k, v, mask_length=k.shape[2], mask="causal"), pc.update_and_fetch(k, v, mask_length=k.shape[2], mask="causal"),
                 f"TinyKvFullCache step {i}")
        rc.rewind(2), pc.rewind(2)
        same(rc.offset, pc.offset, "offset after rewind")
        k, v = chunks[1]
        same(rc.update_and_fetch(k, v), pc.update_and_fetch(k, v), "after rewind")
        # batching cache: requests join and leave, every row has its own length
        rb, pb = R.BatchingKvCache(max_active_requests=3, max_seq_len=32), P.BatchingKvCache(max_active_requests=3, max_seq_len=32)
        for slot, n in ((0, 4), (2, 7)):
            for cache_cls, batch in ((R.TinyKvFullCache, rb), (P.TinyKvFullCache, pb)):
                one = cache_cls()
                one.update_and_fetch(*[t[:, :, :n] for t in (mx.ones((1, 2, 8, 16), mx.bfloat16) * (slot + 1), mx.ones((1, 2, 8, 16), mx.bfloat16) * (slot + 2))])
                batch.add_request(one, slot)
        k = mx.random.normal((3, 2, 1, 16)).astype(mx.bfloat16)
        v = mx.random.normal((3, 2, 1, 16)).astype(mx.bfloat16)
        same(rb.update_and_fetch(k, v, mask_length=1, mask="causal"), pb.update_and_fetch(k, v, mask_length=1, mask="causal"), "BatchingKvCache step")
        rb.remove_request(0), pb.remove_request(0)
        same(rb.update_and_fetch(k, v, mask_length=1, mask="causal"), pb.update_and_fetch(k, v, mask_length=1, mask="causal"), "after a request left")


def pool_counters(pool):
    return {name: getattr(pool, name) for name in ("num_pages", "num_free_pages", "num_live_pages", "reused_page_allocations", "storage_growths",
                                                   "copied_pages_on_growth") if hasattr(pool, name)}


def test_page_pool_and_paged_cache_bookkeeping():
    with mx.stream(mx.cpu):
        mx.random.seed(4)
        rp, pp = R.TinyKvPagedPool(page_size=4), P.TinyKvPagedPool(page_size=4)
        rcs, pcs = [R.TinyKvPagedCache(pool=rp) for _ in range(3)], [P.TinyKvPagedCache(pool=pp) for _ in range(3)]
        script = [(0, 6), (1, 3), (0, 1), (2, 9), (1, 1), (0, 1)]
        for step, (who, n) in enumerate(script):
            k = mx.random.normal((1, 2, n, 8)).astype(mx.bfloat16)
            v = mx.random.normal((1, 2, n, 8)).astype(mx.bfloat16)
            rm, pm = rcs[who].update_and_fetch_paged(k, v), pcs[who].update_and_fetch_paged(k, v)
            for field in ("block_table", "context_lens", "page_size", "key_pages", "value_pages"):
                same(getattr(rm, field), getattr(pm, field), f"step {step}: paged metadata {field}")
            same(pool_counters(rp), pool_counters(pp), f"step {step}: pool counters")
        rcs[2].rewind(5), pcs[2].rewind(5)
        same(pool_counters(rp), pool_counters(pp), "counters after rewind")
        rcs[1].release(), pcs[1].release()
        same(pool_counters(rp), pool_counters(pp), "counters after release")
        k = mx.random.normal((1, 2, 5, 8)).astype(mx.bfloat16)
        rm, pm = rcs[2].update_and_fetch_paged(k, k), pcs[2].update_and_fetch_paged(k, k)
        same(rm.block_table, pm.block_table, "block table after reuse")
        same(pool_counters(rp), pool_counters(pp), "counters after reuse")
        # the dense view of a paged cache (the Week 3 Day 4 compatibility path)
        same(rcs[0].update_and_fetch(k[:, :, :1], k[:, :, :1]), pcs[0].update_and_fetch(k[:, :, :1], k[:, :, :1]), "dense gather of a paged cache")


# ---- models ---------------------------------------------------------------------------------------------------------------
def test_week1_model(checkpoint):
    with mx.stream(mx.cpu):
        tokens = mx.array([prompt_ids(13, 5)], dtype=mx.int32)
        same(R.Qwen3ModelWeek1(checkpoint)(tokens), P.Qwen3ModelWeek1(checkpoint)(tokens), "Week-1 logits")


@pytest.mark.parametrize("name", ["kv-cache", "quantized-matvec", "rmsnorm", "rope", "swiglu", "decode-attention", "simd-matmul", "split-k", None])
def test_week2_model_at_every_checkpoint(checkpoint, name):
    with mx.stream(mx.cpu):
        kwargs = {} if name is None else {"checkpoint": name}
        rm, pm = R.Qwen3ModelWeek2(checkpoint, **kwargs), P.Qwen3ModelWeek2(checkpoint, **kwargs)
        rc, pc = rm.create_kv_cache(), pm.create_kv_cache()
        prompt = prompt_ids(19, 6)
        tokens = mx.array([prompt], dtype=mx.int32)
        for keep in (None, 1):
            pass
        a, b = rm(tokens, 0, rc), pm(tokens, 0, pc)
        same(a, b, f"Week-2 {name}: prefill logits")
        tok, offset = int(a[0, -1].argmax()), len(prompt)
        for step in range(3):
            t = mx.array([[tok]], dtype=mx.int32)
            a, b = rm(t, offset, rc, logits_to_keep=1), pm(t, offset, pc, logits_to_keep=1)
            same(a, b, f"Week-2 {name}: decode step {step}")
            tok, offset = int(a[0, -1].argmax()), offset + 1
        for c in (*rc, *pc):
            c.release()


def moe_checkpoint():
    cfg = dict(TINY_CFG, **MOE_CFG_OVERRIDES)
    with mx.stream(mx.cpu):
        tree = to_mlx_shaped(cfg, O.make_qwen3_weights(cfg, seed=9, sigma=0.05), device="cpu")
    from types import SimpleNamespace as NS

    w = make_moe_weights(cfg, seed=9)
    for layer, lw in zip(tree.model.layers, w["layers"]):
        if "moe" not in lw:
            continue

        def q(triple):
            packed, scales, biases = triple
            return NS(weight=torch.from_numpy(np.ascontiguousarray(packed).view(np.int32)), scales=torch.from_numpy(scales).to(torch.bfloat16),
                      biases=torch.from_numpy(biases).to(torch.bfloat16), group_size=128, bits=4)

        layer.mlp = NS(gate=q(lw["moe"]["router"]), switch_mlp=NS(gate_proj=q(lw["moe"]["gate_proj"]), up_proj=q(lw["moe"]["up_proj"]),
                                                                  down_proj=q(lw["moe"]["down_proj"])))
    return tree


@pytest.mark.parametrize("paged", [True, False], ids=["paged", "dense-gather"])
@pytest.mark.parametrize("moe", [False, True], ids=["dense-mlp", "qwen3-moe"])
def test_week3_model_with_staggered_batching(checkpoint, paged, moe):
    """Three requests joining two steps apart on a BatchingKvCache of per-request paged caches (reference
    tests_refsol/test_week_3_day_1.py:128-195), prefill in chunks first."""
    with mx.stream(mx.cpu):
        tree = moe_checkpoint() if moe else checkpoint
        same = (lambda a, b, what: close(a, b, what, 0.04)) if moe else globals()["same"]  # logits of O(3): see close()
        rm, pm = R.Qwen3ModelWeek3(tree, page_size=8, enable_paged_attention=paged), P.Qwen3ModelWeek3(tree, page_size=8, enable_paged_attention=paged)
        # single request: chunked prefill (two chunks) then decode
        rc, pc = rm.create_kv_cache(), pm.create_kv_cache()
        prompt = prompt_ids(21, 7)
        for start, stop in ((0, 16), (16, 21)):
            t = mx.array([prompt[start:stop]], dtype=mx.int32)
            a, b = rm(t, start, rc, logits_to_keep=1), pm(t, start, pc, logits_to_keep=1)
            same(a, b, f"Week-3 chunk {start}:{stop}")
        t = mx.array([[int(a[0, -1].argmax())]], dtype=mx.int32)
        same(rm(t, 21, rc), pm(t, 21, pc), "Week-3 decode")
        for c in (*rc, *pc):
            c.release()
        # continuous batching
        starts, seq_len = [0, 2, 4], 4
        inputs = np.random.default_rng(8).integers(1, 200, size=(3, seq_len))
        rb = [R.BatchingKvCache(max_active_requests=3, max_seq_len=64) for _ in range(rm.num_hidden_layers)]
        pb = [P.BatchingKvCache(max_active_requests=3, max_seq_len=64) for _ in range(pm.num_hidden_layers)]
        for step in range(seq_len + starts[-1]):
            index = [step - s for s in starts]
            for rid, sidx in enumerate(index):
                if sidx == 0:
                    for batch, own in ((rb, rm.create_kv_cache()), (pb, pm.create_kv_cache())):
                        for layer_batch, layer_own in zip(batch, own):
                            layer_batch.add_request(layer_own, rid)
                elif sidx == seq_len:
                    for layer_batch in (*rb, *pb):
                        layer_batch.remove_request(rid)
            tokens = [int(inputs[r, s]) if 0 <= s < seq_len else 0 for r, s in enumerate(index)]
            offsets = [s if 0 <= s < seq_len else 0 for s in index]
            t, o = mx.array(tokens, dtype=mx.int32).reshape(-1, 1), mx.array(offsets, dtype=mx.int32)
            same(rm(t, o, rb), pm(t, o, pb), f"Week-3 batched step {step}")
        for r_pool, p_pool in zip(rm.page_pools, pm.page_pools):
            globals()["same"](pool_counters(r_pool), pool_counters(p_pool), "page-pool counters after the batch")


# ---- generation loops, scheduler, speculative decoding -------------------------------------------------------------------
@pytest.fixture(scope="module")
def stand_in(tmp_path_factory):
    home = tmp_path_factory.mktemp("hf")
    write_stand_in_checkpoints(home, eos_friendly=True)
    import os

    os.environ["HF_HOME"], os.environ["HF_HUB_OFFLINE"] = str(home), "1"
    from mlx_lm import load

    with mx.stream(mx.cpu):
        return load("Qwen/Qwen3-8B-MLX-4bit")  # greedy generation on it meets <eos> (tests/checkpoint_fixture.py)


def close(a, b, what, atol):
    """For the MoE layers only: the reference's experts run through `mx.gather_qmm` (the facade: fp32 torch matmul), the product's
    through its own grouped-expert primitive (here: the oracle, float64 accumulation) -- two different PRIMITIVES, so the results
    agree to the rounding of one bf16 value per expert output, not bit for bit.  Everything around them is still compared exactly."""
    if a.dtype != b.dtype or tuple(a.shape) != tuple(b.shape):
        raise AssertionError(f"{what}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}")
    worst = float((a.double() - b.double()).abs().max())
    if worst > atol:
        raise AssertionError(f"{what}: max |difference| {worst:.3e} > {atol}")


def same_run(ref, own, what):
    """(result, printed) pairs: the printed text must be identical; the reference's loops return None where the product also hands
    back the text it printed."""
    same(ref[1], own[1], what + ": printed output")
    if ref[0] is not None:
        same(ref[0], own[0], what + ": result")
    else:
        assert own[0] is None or own[0] == own[1], what


def captured(fn, *args, **kwargs):
    """(result, what it printed) with the scheduler's wall-clock stamps ('--- 0:00:00.019935') blanked."""
    import contextlib
    import io
    import re

    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        result = fn(*args, **kwargs)
    return result, re.sub(r"--- \d+:\d\d:\d\d(\.\d+)?", "--- <elapsed>", out.getvalue())


def test_generation_loops(stand_in):
    model, tokenizer = stand_in
    with mx.stream(mx.cpu):
        prompt = "w1 w2 w3 w4"
        # greedy (sampler None); with a sampler the two draw from different random streams -- what their samplers hand to the random
        # primitive is compared in test_moe_block_and_sampler
        ref = captured(R.simple_generate, R.Qwen3ModelWeek1(model), tokenizer, prompt, None)
        own = captured(P.simple_generate, P.Qwen3ModelWeek1(model), tokenizer, prompt, None)
        same_run(ref, own, "simple_generate (Week 1)")
        for week, kwargs in ((2, dict(checkpoint="kv-cache")), (2, dict()), (3, dict())):
            rm = (R.Qwen3ModelWeek2 if week == 2 else R.Qwen3ModelWeek3)(model, **kwargs)
            pm = (P.Qwen3ModelWeek2 if week == 2 else P.Qwen3ModelWeek3)(model, **kwargs)
            same_run(captured(R.simple_generate_with_kv_cache, rm, tokenizer, prompt), captured(P.simple_generate_with_kv_cache, pm, tokenizer, prompt),
                     f"simple_generate_with_kv_cache (Week {week}, {kwargs})")
        for week in (2, 3):
            cls_r, cls_p = (R.Qwen3ModelWeek2, P.Qwen3ModelWeek2) if week == 2 else (R.Qwen3ModelWeek3, P.Qwen3ModelWeek3)
            ref = captured(R.speculative_generate, cls_r(model), cls_r(model), tokenizer, tokenizer, prompt)
            own = captured(P.speculative_generate, cls_p(model), cls_p(model), tokenizer, tokenizer, prompt)
            same_run(ref, own, f"speculative_generate (Week {week})")


@pytest.mark.parametrize("week", [2, 3])
def test_continuous_batching_scheduler(stand_in, week):
    model, tokenizer = stand_in
    prompts = ["w1 w2 w3", "w9 " * 30, "w5", "w7 w8 " * 9, "w3 w3 w3 w3 w3", "w11 w12"]
    with mx.stream(mx.cpu):
        rm = (R.Qwen3ModelWeek2 if week == 2 else R.Qwen3ModelWeek3)(model)
        pm = (P.Qwen3ModelWeek2 if week == 2 else P.Qwen3ModelWeek3)(model)
        ref = captured(R.batch_generate, rm, tokenizer, prompts, max_seq_len=48, batch_size=3, prefill_step=16)
        own = captured(P.batch_generate, pm, tokenizer, prompts, max_seq_len=48, batch_size=3, prefill_step=16)
        same(ref[0], own[0], f"batch_generate results (Week {week})")
        same(ref[1], own[1], f"batch_generate progress output (Week {week})")


def test_moe_block_and_sampler(monkeypatch):
    with mx.stream(mx.cpu):
        tree = moe_checkpoint()
        mlp = tree.model.layers[1].mlp
        args = dict(num_experts_per_tok=2, norm_topk_prob=True)
        rw = {k: R.QuantizedWeights.from_mlx_layer(v) for k, v in (("w_router", mlp.gate), ("w_gate", mlp.switch_mlp.gate_proj),
                                                                 ("w_up", mlp.switch_mlp.up_proj), ("w_down", mlp.switch_mlp.down_proj))}
        pw = {k: P.QuantizedWeights.from_mlx_layer(v) for k, v in (("w_router", mlp.gate), ("w_gate", mlp.switch_mlp.gate_proj),
                                                                 ("w_up", mlp.switch_mlp.up_proj), ("w_down", mlp.switch_mlp.down_proj))}
        mx.random.seed(6)
        x = (mx.random.normal((2, 3, 256)) * 0.5).astype(mx.bfloat16)
        same(R.route_topk(x, rw["w_router"], 2, True), P.route_topk(x, pw["w_router"], 2, True), "route_topk")
        same(R.grouped_expert_linear(x, rw["w_gate"], mx.array([[0, 3, 1], [2, 2, 0]], dtype=mx.int32)).shape,
             P.grouped_expert_linear(x, pw["w_gate"], mx.array([[0, 3, 1], [2, 2, 0]], dtype=mx.int32)).shape, "grouped_expert_linear shape")
        close(R.grouped_expert_linear(x, rw["w_gate"], mx.array([[0, 3, 1], [2, 2, 0]], dtype=mx.int32)),
              P.grouped_expert_linear(x, pw["w_gate"], mx.array([[0, 3, 1], [2, 2, 0]], dtype=mx.int32)), "grouped_expert_linear", 2.0 ** -7)
        close(R.Moe(**rw, **args)(x), P.Moe(**pw, **args)(x), "Moe block", 2.0 ** -7)
        # Samplers: the two draw from different random streams (mx.random.categorical vs torch.multinomial), so what is compared is
        # the DISTRIBUTION each hands to its random primitive -- temperature, top-k and top-p filtering -- on one row, as every
        # caller in the reference has it (its `logprobs[:, mask_elements] = -inf` indexes the columns of ALL rows with every row's
        # index set, so it is only meaningful for a batch of one).
        seen = {}
        monkeypatch.setattr(mx.random, "categorical", lambda logits, axis=-1, **_: seen.__setitem__("ref", torch.softmax(logits.float(), dim=axis)) or
                            torch.zeros(logits.shape[:-1], dtype=torch.int64))
        monkeypatch.setattr(torch, "multinomial", lambda probs, n, **_: seen.__setitem__("own", probs.float()) or
                            torch.zeros((*probs.shape[:-1], n), dtype=torch.int64))
        mx.random.seed(11)
        for row in torch.log_softmax(mx.random.normal((12, 1, 50)) * 2.0, dim=-1):
            for temp, top_p, top_k in ((0.7, None, 5), (1.0, 0.6, None), (0.9, 0.8, 10), (1.3, None, None)):
                seen.clear()
                R.make_sampler(temp, top_p, top_k)(row.clone())
                P.make_sampler(temp, top_p, top_k)(row.clone())
                ref, own = seen["ref"].reshape(-1), seen["own"].reshape(-1)
                same((ref > 0).tolist(), (own > 0).tolist(), f"sampler support temp={temp} top_p={top_p} top_k={top_k}")
                close(ref, own / own.sum(), f"sampler distribution temp={temp} top_p={top_p} top_k={top_k}", 1e-6)
            same(R.make_sampler(0, None, None)(row).tolist(), P.make_sampler(0, None, None)(row).tolist(), "greedy sampler")


# ---- error behaviour --------------------------------------------------------------------------------------------------------
def outcome(fn):
    try:
        fn()
    except Exception as exc:  # noqa: BLE001  (the point is to compare whatever is raised)
        return type(exc).__name__, str(exc)
    return "no error", ""


def test_precondition_failures_read_alike():
    """Every rejected call must be rejected by both code bases with the same exception type and the same message."""
    with mx.stream(mx.cpu):
        mx.random.seed(12)
        P_, Hkv, page, D, Hq = 6, 2, 4, 8, 4
        kp = mx.random.normal((P_, Hkv, page, D)).astype(mx.bfloat16)
        vp = mx.random.normal((P_, Hkv, page, D)).astype(mx.bfloat16)
        q = mx.random.normal((2, Hq, 1, D)).astype(mx.bfloat16)
        table = mx.array([[0, 1, -1], [2, 3, 4]], dtype=mx.int32)
        ctx = mx.array([5, 9], dtype=mx.int32)
        good = dict(query=q, key_pages=kp, value_pages=vp, block_table=table, context_lens=ctx, page_size=page, scale=0.3, mask=None)

        def call(module, **changes):
            args = dict(good, **changes)
            return lambda: module.paged_attention(args["query"], args["key_pages"], args["value_pages"], args["block_table"], args["context_lens"],
                                                  args["page_size"], scale=args["scale"], mask=args["mask"])

        bad = [
            dict(mask=mx.zeros((2, Hq, 1, 9))), dict(mask="full"), dict(query=q[0]), dict(key_pages=kp[0]), dict(value_pages=vp[:, :1]),
            dict(block_table=table[0]), dict(context_lens=ctx[None]), dict(block_table=table.astype(mx.float32)), dict(page_size=0),
            dict(page_size=page + 1), dict(query=mx.random.normal((2, 3, 1, D)).astype(mx.bfloat16)),
            dict(query=mx.random.normal((2, Hq, 1, D + 8)).astype(mx.bfloat16)), dict(context_lens=mx.array([5], dtype=mx.int32)),
            dict(query=q.astype(mx.float32)), dict(query=q.astype(mx.float16), key_pages=kp.astype(mx.float16), value_pages=vp.astype(mx.float16)),
            dict(context_lens=mx.array([-1, 9], dtype=mx.int32)), dict(context_lens=mx.array([5, 13], dtype=mx.int32)),
            dict(block_table=mx.array([[0, 1, -1], [2, 1, 4]], dtype=mx.int32)), dict(block_table=mx.array([[0, 7, -1], [2, 3, 4]], dtype=mx.int32)),
            dict(block_table=mx.array([[0, -1, 1], [2, 3, 4]], dtype=mx.int32)), dict(block_table=mx.zeros((2, 0), dtype=mx.int32)),
        ]
        same(outcome(call(R)), outcome(call(P)), "the valid call")
        assert outcome(call(R))[0] == "no error"
        for changes in bad:
            ref, own = outcome(call(R, **changes)), outcome(call(P, **changes))
            assert ref[0] != "no error", f"the reference accepts {sorted(changes)}"
            same(ref, own, f"paged_attention with {sorted(changes)}")
        # caches and the pool
        for make_r, make_p, what in (
            (lambda: R.TinyKvFullCache().rewind(1), lambda: P.TinyKvFullCache().rewind(1), "rewind past the start of a dense cache"),
            (lambda: R.TinyKvPagedPool(page_size=0), lambda: P.TinyKvPagedPool(page_size=0), "a pool with page size 0"),
            (lambda: R.BatchingKvCache(max_active_requests=1, max_seq_len=8).remove_request(0),
             lambda: P.BatchingKvCache(max_active_requests=1, max_seq_len=8).remove_request(0), "removing a request that was never added"),
            (lambda: R.BatchingKvCache(max_active_requests=1, max_seq_len=8).add_request(R.TinyKvFullCache(), 3),
             lambda: P.BatchingKvCache(max_active_requests=1, max_seq_len=8).add_request(P.TinyKvFullCache(), 3), "adding a request beyond the batch"),
            (lambda: R.quantized_matmul(mx.zeros((2, 1), mx.bfloat16), mx.zeros((2, 1), mx.bfloat16), 64, 4, mx.zeros((1, 128), mx.bfloat16),
                                        mx.zeros((2, 16), mx.uint32), True),
             lambda: P.quantized_matmul(mx.zeros((2, 1), mx.bfloat16), mx.zeros((2, 1), mx.bfloat16), 64, 4, mx.zeros((1, 128), mx.bfloat16),
                                        mx.zeros((2, 16), mx.uint32), True), "a group size the kernel does not support"),
            (lambda: R.dispatch_model("llama-3", None, week=2), lambda: P.dispatch_model("llama-3", None, week=2), "an unknown model family"),
            (lambda: R.make_sampler(0.5, None, None)(mx.zeros((1, 4)) + float("nan")), lambda: P.make_sampler(0.5, None, None)(mx.zeros((1, 4)) + float("nan")),
             "NaN log-probabilities"),
        ):
            ref, own = outcome(make_r), outcome(make_p)
            if ref[0] == "no error":
                assert own[0] == "no error", f"{what}: the reference accepts it, the product raises {own}"
            else:
                assert own[0] == ref[0], f"{what}: {ref} vs {own}"


# ---- the bench harness's pure-Python parts ---------------------------------------------------------------------------------
def test_bench_trace_generator_and_percentiles():
    """This repository's benches/bench.py and benches/serving.py against the reference's benches/bench.py: the seeded request trace
    (reference build_requests, 201-225) and the report's order statistics (sample_median 575-576, nearest_rank_percentile 579-585)."""
    import importlib.util
    from random import Random

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        module = importlib.util.module_from_spec(spec)
        sys.modules[name] = module
        spec.loader.exec_module(module)
        return module

    ref = load("reference_benches_bench", REFERENCE / "benches" / "bench.py")
    own = load("product_benches_bench", ROOT / "benches" / "bench.py")
    serving = load("product_benches_serving", ROOT / "benches" / "serving.py")
    for seed, vocab, eos in ((0, 151936, 151645), (7, 1024, 0), (3, 300, 299)):
        kwargs = dict(num_seqs=9, vocab_size=vocab, eos_token_id=eos, min_input_len=3, max_input_len=40, min_output_len=1, max_output_len=9)
        a, b = ref.build_requests(rng=Random(seed), **kwargs), own.build_requests(rng=Random(seed), **kwargs)
        same([(r.prompt_token_ids, r.max_new_tokens) for r in a], [(r.prompt_token_ids, r.max_new_tokens) for r in b], f"trace, seed {seed}")
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 20, 101):
        samples = [float(x) for x in rng.random(n)]
        same(ref.sample_median(samples), serving.median(samples), f"median of {n}")
        for q in (0.5, 0.9, 0.95, 0.99, 1.0):
            same(ref.nearest_rank_percentile(samples, q), serving.nearest_rank(samples, q), f"p{q} of {n}")
    same(ref.safe_div(3.0, 0.0), own.safe_div(3.0, 0.0), "safe_div by zero")
