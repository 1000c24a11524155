"""`mlx_lm.models.qwen3` layers as the reference's Week-1 tests use them: ORACLES for the course's own attention /
MLP / transformer block (tests_refsol/test_week_1_day_3.py:149-199, test_week_1_day_4.py:106-112, test_week_1_day_5.py:14-80).

Restated in fp32 torch over the `mlx.nn` facade from the published mlx-lm 0.31 semantics:
bias-free projections, per-head RMSNorm of q and k, non-traditional RoPE over the whole head, lower-right causal mask,
pre-norm residual block.  PARITY UNPINNED against real mlx-lm (not installable in this image).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import mlx.core as mx
import mlx.nn as nn


@dataclass
class ModelArgs:
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    num_key_value_heads: int
    max_position_embeddings: int
    rope_theta: float
    head_dim: int
    tie_word_embeddings: bool
    rope_scaling: Optional[dict] = None

    @classmethod
    def from_dict(cls, params: dict):
        names = cls.__dataclass_fields__
        return cls(**{k: v for k, v in params.items() if k in names})


class _Rope:
    def __init__(self, dims: int, base: float, traditional: bool = False):
        self.dims, self.base, self.traditional = dims, base, traditional

    def __call__(self, x, offset: Any = 0):
        return mx.fast.rope(x, self.dims, traditional=self.traditional, base=self.base, scale=1.0, offset=offset)


class Attention(nn.Module):
    def __init__(self, args: ModelArgs):
        dim = args.hidden_size
        self.n_heads = args.num_attention_heads
        self.n_kv_heads = args.num_key_value_heads
        head_dim = args.head_dim
        self.scale = head_dim ** -0.5
        self.q_proj = nn.Linear(dim, self.n_heads * head_dim, bias=False)
        self.k_proj = nn.Linear(dim, self.n_kv_heads * head_dim, bias=False)
        self.v_proj = nn.Linear(dim, self.n_kv_heads * head_dim, bias=False)
        self.o_proj = nn.Linear(self.n_heads * head_dim, dim, bias=False)
        self.q_norm = nn.RMSNorm(head_dim, eps=args.rms_norm_eps)
        self.k_norm = nn.RMSNorm(head_dim, eps=args.rms_norm_eps)
        self.rope = _Rope(head_dim, base=args.rope_theta, traditional=False)

    def __call__(self, x, mask=None, cache=None):
        B, L, _ = x.shape
        q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        q = self.q_norm(q.reshape(B, L, self.n_heads, -1)).permute(0, 2, 1, 3)
        k = self.k_norm(k.reshape(B, L, self.n_kv_heads, -1)).permute(0, 2, 1, 3)
        v = v.reshape(B, L, self.n_kv_heads, -1).permute(0, 2, 1, 3)
        if cache is not None:
            q = self.rope(q, offset=cache.offset)
            k = self.rope(k, offset=cache.offset)
            k, v = cache.update_and_fetch(k, v)
        else:
            q, k = self.rope(q), self.rope(k)
        out = mx.fast.scaled_dot_product_attention(q, k, v, scale=self.scale, mask=mask)
        return self.o_proj(out.permute(0, 2, 1, 3).reshape(B, L, -1))


class MLP(nn.Module):
    def __init__(self, dim: int, hidden_dim: int):
        self.gate_proj = nn.Linear(dim, hidden_dim, bias=False)
        self.down_proj = nn.Linear(hidden_dim, dim, bias=False)
        self.up_proj = nn.Linear(dim, hidden_dim, bias=False)

    def __call__(self, x):
        return self.down_proj(nn.silu(self.gate_proj(x)) * self.up_proj(x))


class TransformerBlock(nn.Module):
    def __init__(self, args: ModelArgs):
        self.num_attention_heads = args.num_attention_heads
        self.hidden_size = args.hidden_size
        self.self_attn = Attention(args)
        self.mlp = MLP(args.hidden_size, args.intermediate_size)
        self.input_layernorm = nn.RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.post_attention_layernorm = nn.RMSNorm(args.hidden_size, eps=args.rms_norm_eps)
        self.args = args

    def __call__(self, x, mask=None, cache=None):
        h = x + self.self_attn(self.input_layernorm(x), mask, cache)
        return h + self.mlp(self.post_attention_layernorm(h))


class Qwen3Model(nn.Module):
    def __init__(self, args: ModelArgs):
        self.args = args
        self.vocab_size = args.vocab_size
        self.num_hidden_layers = args.num_hidden_layers
        self.embed_tokens = None  # set by from_checkpoint (a W4 table; the reference only ever loads 4-bit exports)
        self.layers = [TransformerBlock(args) for _ in range(args.num_hidden_layers)]
        self.norm = nn.RMSNorm(args.hidden_size, eps=args.rms_norm_eps)

    def __call__(self, inputs, cache=None, input_embeddings=None):
        h = input_embeddings if input_embeddings is not None else self.embed_tokens(inputs)
        if cache is None:
            cache = [None] * len(self.layers)
        mask = "causal" if h.shape[1] > 1 else None  # lower-right aligned over the cached context (mlx_lm create_attention_mask)
        for layer, c in zip(self.layers, cache):
            h = layer(h, mask, c)
        return self.norm(h)


class Model(nn.Module):
    """The object `mlx_lm.load` returns for a Qwen3 checkpoint: callable ([B, L] ids -> [B, L, V] logits) AND the attribute
    tree the course models read their weights from (`.args`, `.model.embed_tokens`, `.model.layers[i].self_attn.q_proj.
    {weight, scales, biases, group_size, bits}` ..., reference qwen3_week2.py:288-350).  The reference's model-level tests use
    the call as their oracle (tests_refsol/test_week_1_day_5.py:110-123, test_week_2_day_6.py:124-148, test_week_3_day_1.py:
    150-195) and benches/bench.py:315-348 times it as `--solution mlx`."""

    def __init__(self, args: ModelArgs):
        self.args = args
        self.model_type = args.model_type
        self.model = Qwen3Model(args)
        if not args.tie_word_embeddings:
            self.lm_head = None  # set by from_checkpoint

    def __call__(self, inputs, cache=None, input_embeddings=None):
        out = self.model(inputs, cache, input_embeddings)
        if self.args.tie_word_embeddings:
            return self.model.embed_tokens.as_linear(out)
        return self.lm_head(out)

    @property
    def layers(self):
        return self.model.layers

    @classmethod
    def from_checkpoint(cls, tree):
        """Wrap the loader's weight tree (tiny_llm_hip/loader.py:load_weights, a SimpleNamespace shaped like an mlx_lm
        model) into callable facade layers that share its tensors."""
        a = tree.args
        args = ModelArgs(model_type="qwen3", hidden_size=a.hidden_size, num_hidden_layers=a.num_hidden_layers,
                         intermediate_size=a.intermediate_size, num_attention_heads=a.num_attention_heads,
                         rms_norm_eps=a.rms_norm_eps, vocab_size=a.vocab_size, num_key_value_heads=a.num_key_value_heads,
                         max_position_embeddings=getattr(a, "max_position_embeddings", 40960), rope_theta=a.rope_theta,
                         head_dim=a.head_dim, tie_word_embeddings=a.tie_word_embeddings)
        for key, value in vars(a).items():  # fields beyond the dense Qwen3 arguments (num_experts, norm_topk_prob, ...)
            if not hasattr(args, key):
                setattr(args, key, value)
        model = cls(args)

        def q(layer):
            return nn.QuantizedLinear(layer.weight, layer.scales, layer.biases, layer.group_size, layer.bits)

        e = tree.model.embed_tokens
        model.model.embed_tokens = nn.QuantizedEmbedding(e.weight, e.scales, e.biases, e.group_size, e.bits)
        for block, src in zip(model.model.layers, tree.model.layers):
            for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(block.self_attn, name, q(getattr(src.self_attn, name)))
            block.self_attn.q_norm.weight = src.self_attn.q_norm.weight
            block.self_attn.k_norm.weight = src.self_attn.k_norm.weight
            if hasattr(src.mlp, "switch_mlp"):  # Qwen3-MoE layer (mlx_lm.models.qwen3_moe): router + grouped experts
                from .qwen3_moe import Qwen3MoeSparseMoeBlock
                from .switch_layers import QuantizedSwitchLinear

                moe = Qwen3MoeSparseMoeBlock(a)
                moe.gate = q(src.mlp.gate)
                for name in ("gate_proj", "up_proj", "down_proj"):
                    e = getattr(src.mlp.switch_mlp, name)
                    setattr(moe.switch_mlp, name, QuantizedSwitchLinear(e.weight, e.scales, e.biases, e.group_size, e.bits))
                block.mlp = moe
            else:
                for name in ("gate_proj", "up_proj", "down_proj"):
                    setattr(block.mlp, name, q(getattr(src.mlp, name)))
            block.input_layernorm.weight = src.input_layernorm.weight
            block.post_attention_layernorm.weight = src.post_attention_layernorm.weight
        model.model.norm.weight = tree.model.norm.weight
        if not args.tie_word_embeddings:
            model.lm_head = q(tree.lm_head)
        return model
