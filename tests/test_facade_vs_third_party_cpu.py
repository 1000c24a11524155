"""CPU tier: the facade's `mx.fast.*`, `mlx.nn` and `mlx_lm` layers -- the ORACLES the reference's own tests compare the
course code with (tests/test_refsol_facade_cpu.py) -- against third-party implementations of the same operators:
PyTorch's own `F.rms_norm`, `F.scaled_dot_product_attention(enable_gqa=...)`, `nn.MultiheadAttention`, complex-number
rotation for the interleaved RoPE, and Hugging Face transformers' Qwen3 attention / MLP / decoder layer / rotary embedding.
The facade restates MLX from its documentation (parity unpinned against MLX itself, compat/README.md); this file shows the
restatement computes what the rest of the ecosystem computes under the same names."""

import sys
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
for extra in (ROOT / "tiny-llm_amd" / "compat", ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(extra) not in sys.path:
        sys.path.insert(0, str(extra))


def mx_nn():
    import mlx.core as mx
    import mlx.nn as nn

    return mx, nn


def test_fast_rms_norm_and_sdpa_against_torch_builtins():
    mx, _ = mx_nn()
    g = torch.Generator().manual_seed(0)
    with mx.stream(mx.cpu):
        x = torch.randn((3, 7, 96), generator=g, dtype=torch.float64).float()
        w = 1 + 0.1 * torch.randn((96,), generator=g)
        torch.testing.assert_close(mx.fast.rms_norm(x, w, 1e-5), F.rms_norm(x, (96,), w, 1e-5), rtol=1e-5, atol=1e-6)
        # grouped-query attention, no mask / causal (lower-right aligned when L < S) / additive mask / boolean mask
        q = torch.randn((2, 8, 5, 32), generator=g)
        k = torch.randn((2, 2, 9, 32), generator=g)
        v = torch.randn((2, 2, 9, 32), generator=g)
        scale = 32 ** -0.5
        want = F.scaled_dot_product_attention(q, k, v, scale=scale, enable_gqa=True)
        torch.testing.assert_close(mx.fast.scaled_dot_product_attention(q, k, v, scale=scale), want, rtol=1e-5, atol=1e-5)
        keep = torch.ones((5, 9), dtype=torch.bool).tril(diagonal=9 - 5)  # query i sees keys 0 .. S - L + i
        want = F.scaled_dot_product_attention(q, k, v, attn_mask=keep, scale=scale, enable_gqa=True)
        torch.testing.assert_close(mx.fast.scaled_dot_product_attention(q, k, v, scale=scale, mask="causal"), want, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(mx.fast.scaled_dot_product_attention(q, k, v, scale=scale, mask=keep), want, rtol=1e-5, atol=1e-5)
        additive = torch.randn((5, 9), generator=g)
        want = F.scaled_dot_product_attention(q, k, v, attn_mask=additive, scale=scale, enable_gqa=True)
        torch.testing.assert_close(mx.fast.scaled_dot_product_attention(q, k, v, scale=scale, mask=additive), want, rtol=1e-5, atol=1e-5)


def test_fast_rope_against_complex_rotation_and_transformers():
    mx, _ = mx_nn()
    transformers = pytest.importorskip("transformers")
    from transformers.models.qwen3.modeling_qwen3 import Qwen3RotaryEmbedding, apply_rotary_pos_emb

    g = torch.Generator().manual_seed(1)
    B, H, L, D, base, offset = 2, 3, 6, 64, 10000.0, 11
    with mx.stream(mx.cpu):
        x = torch.randn((B, H, L, D), generator=g)
        # traditional = interleaved pairs (2d, 2d+1): multiplication by e^{i * pos * base^(-d / (D/2))} on complex pairs
        pos = torch.arange(offset, offset + L, dtype=torch.float64)
        inv = base ** (-torch.arange(D // 2, dtype=torch.float64) / (D // 2))
        rot = torch.polar(torch.ones((L, D // 2), dtype=torch.float64), pos[:, None] * inv[None, :])
        want = torch.view_as_real(torch.view_as_complex(x.double().reshape(B, H, L, D // 2, 2)) * rot).reshape(B, H, L, D)
        got = mx.fast.rope(x, D, traditional=True, base=base, scale=1.0, offset=offset)
        torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5)
        # non-traditional = halves (d, d + D/2): transformers' rotate_half convention, cos / sin from its rotary module
        cfg = transformers.Qwen3Config(hidden_size=H * D, num_attention_heads=H, num_key_value_heads=H, head_dim=D, rope_theta=base,
                                       max_position_embeddings=128)
        cos, sin = Qwen3RotaryEmbedding(cfg)(x, torch.arange(offset, offset + L)[None, :].expand(B, L))
        want, _ = apply_rotary_pos_emb(x, x, cos, sin)
        got = mx.fast.rope(x, D, traditional=False, base=base, scale=1.0, offset=offset)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
        # per-row offsets (the batched decode of Week 3) = the same rotation with each row's own positions
        offsets = torch.tensor([3, 40], dtype=torch.int32)
        cos, sin = Qwen3RotaryEmbedding(cfg)(x, offsets[:, None].long() + torch.arange(L)[None, :])
        want, _ = apply_rotary_pos_emb(x, x, cos, sin)
        torch.testing.assert_close(mx.fast.rope(x, D, traditional=False, base=base, scale=1.0, offset=offsets), want, rtol=1e-5, atol=1e-5)


def test_mlx_nn_multi_head_attention_against_torch_nn():
    mx, nn = mx_nn()
    g = torch.Generator().manual_seed(2)
    with mx.stream(mx.cpu):
        E, Hn, L = 48, 4, 7
        ours = nn.MultiHeadAttention(E, Hn)
        theirs = torch.nn.MultiheadAttention(E, Hn, bias=False, batch_first=True)
        with torch.no_grad():
            theirs.in_proj_weight.copy_(torch.cat([ours.query_proj.weight, ours.key_proj.weight, ours.value_proj.weight]))
            theirs.out_proj.weight.copy_(ours.out_proj.weight)
        q, k, v = (torch.randn((3, L, E), generator=g) for _ in range(3))
        mask = torch.randn((L, L), generator=g)
        want, _ = theirs(q, k, v, attn_mask=mask, need_weights=False)
        torch.testing.assert_close(ours(q, k, v, mask=mask), want, rtol=1e-4, atol=1e-5)
        want, _ = theirs(q, k, v, need_weights=False)
        torch.testing.assert_close(ours(q, k, v), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mask", [None, "causal"])
def test_mlx_lm_qwen3_layers_against_transformers(mask):
    """The three oracles of the reference's Week-1 model tests (tests_refsol/test_week_1_day_3.py:149-199, day_4:106-112,
    day_5:14-80): mlx_lm.models.qwen3.{Attention, MLP, TransformerBlock} as the facade restates them."""
    mx, _ = mx_nn()
    transformers = pytest.importorskip("transformers")
    from mlx_lm.models import qwen3
    from transformers.models.qwen3 import modeling_qwen3 as hf

    with mx.stream(mx.cpu):
        mx.random.seed(3)
        args = qwen3.ModelArgs(model_type="qwen3", hidden_size=64, num_hidden_layers=1, intermediate_size=160,
                               num_attention_heads=4, num_key_value_heads=2, head_dim=32, rms_norm_eps=1e-6, vocab_size=100,
                               max_position_embeddings=64, rope_theta=10000, tie_word_embeddings=True)
        block = qwen3.TransformerBlock(args)
        for norm in (block.input_layernorm, block.post_attention_layernorm, block.self_attn.q_norm, block.self_attn.k_norm):
            norm.weight = 1 + 0.1 * mx.random.normal(norm.weight.shape)
        cfg = transformers.Qwen3Config(vocab_size=100, hidden_size=64, intermediate_size=160, num_hidden_layers=1,
                                       num_attention_heads=4, num_key_value_heads=2, head_dim=32, rms_norm_eps=1e-6, rope_theta=10000,
                                       max_position_embeddings=64, attention_bias=False, use_sliding_window=False)
        cfg._attn_implementation = "eager"
        layer = hf.Qwen3DecoderLayer(cfg, layer_idx=0).eval()
        with torch.no_grad():
            for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
                getattr(layer.self_attn, name).weight.copy_(getattr(block.self_attn, name).weight)
            layer.self_attn.q_norm.weight.copy_(block.self_attn.q_norm.weight)
            layer.self_attn.k_norm.weight.copy_(block.self_attn.k_norm.weight)
            for name in ("gate_proj", "up_proj", "down_proj"):
                getattr(layer.mlp, name).weight.copy_(getattr(block.mlp, name).weight)
            layer.input_layernorm.weight.copy_(block.input_layernorm.weight)
            layer.post_attention_layernorm.weight.copy_(block.post_attention_layernorm.weight)
        B, L = 2, 9
        x = mx.random.uniform(-1.0, 1.0, shape=(B, L, 64))
        pos = torch.arange(L)[None, :].expand(B, L)
        cos_sin = hf.Qwen3RotaryEmbedding(cfg)(x, pos)
        additive = None
        if mask == "causal":
            additive = torch.full((L, L), float("-inf")).triu(diagonal=1)[None, None]
        with torch.no_grad():
            want_attn, _ = layer.self_attn(hidden_states=x, position_embeddings=cos_sin, attention_mask=additive)
            want_mlp = layer.mlp(x)
            want_block = layer(x, attention_mask=additive, position_ids=pos, position_embeddings=cos_sin)
        want_block = want_block[0] if isinstance(want_block, tuple) else want_block
        torch.testing.assert_close(block.self_attn(x, mask=mask, cache=None), want_attn, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(block.mlp(x), want_mlp, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(block(x, mask=mask, cache=None), want_block, rtol=1e-4, atol=1e-5)
