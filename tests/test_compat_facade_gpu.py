"""GPU tier: MLX-style code -- `import mlx.core as mx`, `from tiny_llm_ref import ...` -- running on the HIP kernels through
the import facade (tiny-llm_amd/compat).  The reference's own test files cannot travel to the GPU box (/root/reference exists
only in the build container; there they run unmodified against the facade with the oracle standing in for the kernels,
tests/test_refsol_facade_cpu.py), so this file repeats their PATTERNS (own wording) on the real device: mx-built inputs,
the course operator against the `mx.*` built-in as oracle, the reference's tolerances (tests_refsol/utils.py:72-110:
bf16 rtol 5e-2 / atol 1e-2; test_week_2_day_3.py, test_week_2_day_4.py, test_week_3_day_1.py, test_week_3_day_4.py).
"""

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
COMPAT = str(ROOT / "tiny-llm_amd" / "compat")
if COMPAT not in sys.path:
    sys.path.insert(0, COMPAT)

pytestmark = pytest.mark.gpu


def close(a, b, rtol=5e-2, atol=1e-2):
    a, b = np.array(a.astype(_mx().float32)), np.array(b.astype(_mx().float32))
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _mx():
    import mlx.core as mx

    return mx


def test_facade_arrays_are_device_tensors_with_mlx_method_names():
    mx = _mx()
    import torch

    with mx.stream(mx.gpu):
        a = mx.random.normal((4, 6)).astype(mx.bfloat16)
        assert isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == mx.bfloat16
        assert a.transpose(1, 0).shape == (6, 4) and mx.zeros((2, 3, 5)).transpose(2, 0, 1).shape == (5, 2, 3)
        b = mx.array([1, 2, 3], dtype=mx.int32)
        assert b.at[1].add(5).tolist() == [1, 7, 3] and b.tolist() == [1, 2, 3]
        assert np.array(a).shape == (4, 6)  # __array__ of a device tensor
        mx.eval(a, b)
    with mx.stream(mx.cpu):
        assert not mx.zeros((2,)).is_cuda


@pytest.mark.parametrize("rows", [1, 8])
def test_quantized_matvec_against_mx_quantized_matmul(rows):
    """tests_refsol/test_week_2_day_3.py pattern: course kernel vs mx.quantized_matmul on the same packed tensors."""
    mx = _mx()
    from tiny_llm_ref import quantized_matmul

    with mx.stream(mx.gpu):
        mx.random.seed(7)
        weight = (mx.random.normal((2560, 4096)) * 0.05).astype(mx.bfloat16)
        packed, scales, biases = mx.quantize(weight, group_size=128, bits=4)
        x = mx.random.normal((rows, 4096)).astype(mx.bfloat16)
        got = quantized_matmul(scales, biases, 128, 4, x, packed, transpose_b=True, use_simdgroup=True)
        want = mx.quantized_matmul(x, packed, scales, biases, transpose=True, group_size=128, bits=4)
        assert got.dtype == mx.bfloat16 and got.shape == (rows, 2560)
        close(got, want, rtol=5e-2, atol=5e-2)


def test_fast_kernels_against_mx_fast():
    """tests_refsol/test_week_2_day_4.py / test_week_3_day_1.py pattern: FastRMSNorm, FastRoPE (per-row offsets), swiglu."""
    mx = _mx()
    from tiny_llm_ref import FastRMSNorm, FastRoPE, swiglu
    import mlx.nn as nn

    with mx.stream(mx.gpu):
        mx.random.seed(3)
        x = mx.random.normal((2, 5, 256)).astype(mx.bfloat16)
        w = (1.0 + 0.1 * mx.random.normal((256,))).astype(mx.bfloat16)
        close(FastRMSNorm(256, w, eps=1e-6)(x), mx.fast.rms_norm(x, w, 1e-6))
        q = mx.random.normal((3, 4, 8, 128)).astype(mx.bfloat16)  # [B, L, H, D]
        offsets = mx.array([0, 17, 900], dtype=mx.int32)
        for traditional in (False, True):
            got = FastRoPE(128, 4096, 1000000, traditional=traditional)(q, offsets)
            want = mx.fast.rope(q.transpose(0, 2, 1, 3), 128, traditional=traditional, base=1000000, scale=1.0,
                                offset=offsets).transpose(0, 2, 1, 3)
            close(got, want, atol=2e-2)
        g, u = mx.random.normal((7, 512)).astype(mx.bfloat16), mx.random.normal((7, 512)).astype(mx.bfloat16)
        close(swiglu(g, u), (nn.silu(g.astype(mx.float32)) * u.astype(mx.float32)).astype(mx.bfloat16))


def test_paged_attention_against_dense_mx_attention():
    """tests_refsol/test_week_3_day_4.py pattern: paged cache + paged_attention == mx.fast.scaled_dot_product_attention on
    the dense tensors (bench shape 32/8 heads, D 128; reference published max abs err 0.0044 at ctx 128)."""
    mx = _mx()
    from tiny_llm_ref import TinyKvPagedCache, TinyKvPagedPool, paged_attention

    with mx.stream(mx.gpu):
        mx.random.seed(11)
        ctx = 300
        q = mx.random.normal((1, 32, 1, 128)).astype(mx.bfloat16)
        k = mx.random.normal((1, 8, ctx, 128)).astype(mx.bfloat16)
        v = mx.random.normal((1, 8, ctx, 128)).astype(mx.bfloat16)
        pool = TinyKvPagedPool(page_size=128)
        cache = TinyKvPagedCache(pool=pool)
        cache.update_and_fetch(k[:, :, :-1], v[:, :, :-1])
        meta = cache.update_and_fetch_paged(k[:, :, -1:], v[:, :, -1:])
        got = paged_attention(q, meta.key_pages, meta.value_pages, meta.block_table, meta.context_lens, meta.page_size,
                              scale=128 ** -0.5)
        want = mx.fast.scaled_dot_product_attention(q, k, v, scale=128 ** -0.5)
        assert float(mx.max(mx.abs(got.astype(mx.float32) - want.astype(mx.float32))).item()) < 2 * 0.0044
        cache.release()
        assert pool.num_free_pages == pool.num_pages


def test_week3_model_matches_week2_model_on_an_mx_built_checkpoint():
    """tests_refsol/test_week_3_day_3.py:386-402 pattern: Qwen3ModelWeek3 (paged) vs Qwen3ModelWeek2 on a tiny MLX-shaped
    model built with mx.quantize, prefill + incremental decode; the reference asserts rtol = atol = 1e-3 on ITS fake model
    between two paths sharing every kernel -- here the two paths differ in the attention kernel (paged vs dense decode), the
    logits are O(1) bf16 (ulp 2^-8 .. 2^-7), and the band is two ulps."""
    mx = _mx()
    from types import SimpleNamespace as NS

    from tiny_llm_ref import Qwen3ModelWeek2, Qwen3ModelWeek3

    with mx.stream(mx.gpu):
        mx.random.seed(5)

        def ql(o, i):
            packed, scales, biases = mx.quantize((mx.random.normal((o, i)) * 0.08).astype(mx.bfloat16), group_size=128, bits=4)
            return NS(weight=packed, scales=scales, biases=biases, group_size=128, bits=4)

        ones = lambda n: NS(weight=mx.ones((n,), mx.bfloat16))
        args = NS(num_hidden_layers=2, hidden_size=128, vocab_size=256, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                  intermediate_size=256, rms_norm_eps=1e-5, max_position_embeddings=512, rope_theta=10000, tie_word_embeddings=True)
        layers = [NS(self_attn=NS(q_proj=ql(256, 128), k_proj=ql(128, 128), v_proj=ql(128, 128), o_proj=ql(128, 256),
                                  q_norm=ones(128), k_norm=ones(128)),
                     mlp=NS(gate_proj=ql(256, 128), up_proj=ql(256, 128), down_proj=ql(128, 256)),
                     input_layernorm=ones(128), post_attention_layernorm=ones(128)) for _ in range(2)]
        model = NS(args=args, model=NS(embed_tokens=ql(256, 128), layers=layers, norm=ones(128)))
        tokens = mx.array([[5, 17, 200, 3, 99, 41]], dtype=mx.int32)
        outs = []
        for cls in (Qwen3ModelWeek2, Qwen3ModelWeek3):
            net = cls(model)
            cache = net.create_kv_cache()
            logits = [net(tokens, 0, cache, logits_to_keep=1)]
            nxt = mx.array([[7]], dtype=mx.int32)
            logits.append(net(nxt, 6, cache, logits_to_keep=1))
            outs.append(mx.concatenate(logits, axis=1).astype(mx.float32))
            for c in cache:
                c.release()
        assert float(mx.max(mx.abs(outs[0] - outs[1])).item()) <= 2 * 2 ** -7 * max(1.0, float(mx.max(mx.abs(outs[0])).item()))



def _course_fake_model(mx):
    """The fake model of tests_refsol/test_week_3_day_3.py:29-105, restated: two layers, hidden 128, vocabulary 128, four query
    heads over two KV heads of 32, intermediate 256, tied head; N(0, 1) bf16 weights through mx.quantize (group 128, 4 bit),
    unit norm weights; seed 0."""
    from types import SimpleNamespace as NS

    mx.random.seed(0)

    def ql(rows, cols):
        packed, scales, biases = mx.quantize(mx.random.normal(shape=(rows, cols), dtype=mx.bfloat16), group_size=128, bits=4)
        return NS(weight=packed, scales=scales, biases=biases, group_size=128, bits=4)

    ones = lambda n: NS(weight=mx.ones((n,), dtype=mx.bfloat16))
    args = NS(num_hidden_layers=2, hidden_size=128, vocab_size=128, num_attention_heads=4, num_key_value_heads=2, head_dim=32,
              intermediate_size=256, rms_norm_eps=1e-5, max_position_embeddings=128, rope_theta=10000, tie_word_embeddings=True)
    embed = ql(128, 128)
    layers = [NS(self_attn=NS(q_proj=ql(128, 128), k_proj=ql(64, 128), v_proj=ql(64, 128), o_proj=ql(128, 128), q_norm=ones(32),
                              k_norm=ones(32)),
                 mlp=NS(gate_proj=ql(256, 128), up_proj=ql(256, 128), down_proj=ql(128, 256)), input_layernorm=ones(128),
                 post_attention_layernorm=ones(128)) for _ in range(2)]
    return NS(args=args, model=NS(embed_tokens=embed, layers=layers, norm=ones(128)))


@pytest.mark.parametrize("paged", [False, True])
def test_the_reference_1e_3_check_of_week3_against_week2_at_its_own_tolerance(paged):
    """north_star: "logits match ... within 1e-3".  The reference asserts 1e-3 in ONE place -- tests_refsol/test_week_3_day_3.py:
    386-402, Qwen3ModelWeek3(page_size=4, enable_paged_attention=False) against Qwen3ModelWeek2 on its fake model, six tokens fed
    one at a time, log-softmax of the logits, rtol = atol = 1e-3 (both paths share every kernel).  Reproduced here ON THE DEVICE
    at that tolerance, every kernel call reaching libtinyllm_hip.so.  paged=True is NOT a reference assertion: the same loop with
    the paged attention kernel in Week 3 (a different kernel from Week 2's dense decode attention), held at the same 1e-3 and
    reported in profiles/ (the maximum is printed)."""
    mx = _mx()
    from tiny_llm_ref import Qwen3ModelWeek2, Qwen3ModelWeek3

    with mx.stream(mx.gpu):
        fake = _course_fake_model(mx)
        week2 = Qwen3ModelWeek2(fake)
        week3 = Qwen3ModelWeek3(fake, page_size=4, enable_paged_attention=paged)
        tokens = mx.array([[1, 5, 7, 3, 9, 11]], dtype=mx.int32)
        cache2, cache3 = week2.create_kv_cache(), week3.create_kv_cache()
        worst = 0.0
        for offset in range(tokens.shape[1]):
            tok = tokens[:, offset:offset + 1]
            a = week2(tok, offset, cache2)
            b = week3(tok, offset, cache3)
            assert a.is_cuda and b.is_cuda
            a = np.array((a - mx.logsumexp(a, keepdims=True)).astype(mx.float32))
            b = np.array((b - mx.logsumexp(b, keepdims=True)).astype(mx.float32))
            worst = max(worst, float(np.max(np.abs(a - b))))
            assert np.allclose(b, a, rtol=1e-3, atol=1e-3), (offset, worst)
        print(f"PARITY week3_vs_week2_logsoftmax paged={paged} max_abs_diff={worst:.3e} (reference tolerance rtol=atol=1e-3)")
