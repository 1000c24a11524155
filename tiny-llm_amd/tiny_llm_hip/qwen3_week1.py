"""Week-1 Qwen3: dense bf16 weights, readable ops, no KV cache (reference: src/tiny_llm_ref/qwen3_week1.py)."""

from typing import Any

import torch

from .attention import scaled_dot_product_attention_grouped
from .basics import linear, silu
from .embedding import Embedding
from .layer_norm import RMSNorm
from .positional_encoding import RoPE
from .quantize import dequantize_linear


class Qwen3MultiHeadAttention:
    def __init__(self, hidden_size, num_heads, num_kv_heads, head_dim, wq, wk, wv, wo, q_norm, k_norm,
                 max_seq_len: int = 32768, theta: int = 1000000, rms_norm_eps: float = 1e-5):
        assert num_heads % num_kv_heads == 0, (
            f"num_heads {num_heads} must be divisible by num_kv_heads {num_kv_heads}")
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.scale = head_dim ** -0.5
        self.wq, self.wk, self.wv, self.wo = wq, wk, wv, wo
        self.rope = RoPE(head_dim, max_seq_len, theta)
        self.q_norm = RMSNorm(head_dim, q_norm, eps=rms_norm_eps)
        self.k_norm = RMSNorm(head_dim, k_norm, eps=rms_norm_eps)

    def __call__(self, x: torch.Tensor, mask: torch.Tensor | str | None = None) -> torch.Tensor:
        B, L, _ = x.shape
        q = self.q_norm(linear(x, self.wq).reshape(B, L, self.num_heads, self.head_dim))
        k = self.k_norm(linear(x, self.wk).reshape(B, L, self.num_kv_heads, self.head_dim))
        v = linear(x, self.wv).reshape(B, L, self.num_kv_heads, self.head_dim)
        q = self.rope(q, offset=slice(0, L)).transpose(1, 2)
        k = self.rope(k, offset=slice(0, L)).transpose(1, 2)
        v = v.transpose(1, 2)
        # attention itself is evaluated in fp32 (reference qwen3_week1.py:64-70)
        mixed = scaled_dot_product_attention_grouped(
            q.to(torch.float32), k.to(torch.float32), v.to(torch.float32), scale=self.scale, mask=mask
        ).to(x.dtype)
        return linear(mixed.transpose(1, 2).reshape(B, L, self.num_heads * self.head_dim), self.wo)


class Qwen3MLP:
    def __init__(self, dim: int, hidden_dim: int, w_gate, w_up, w_down):
        self.dim = dim
        self.hidden_dim = hidden_dim
        self.w_gate, self.w_up, self.w_down = w_gate, w_up, w_down

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return linear(silu(linear(x, self.w_gate)) * linear(x, self.w_up), self.w_down)


class Qwen3TransformerBlock:
    def __init__(self, num_attention_heads, num_kv_heads, hidden_size, head_dim, intermediate_size, rms_norm_eps,
                 wq, wk, wv, wo, q_norm, k_norm, w_gate, w_up, w_down, w_input_layernorm,
                 w_post_attention_layernorm, max_seq_len: int = 32768, theta: int = 1000000):
        self.mlp = Qwen3MLP(hidden_size, intermediate_size, w_gate, w_up, w_down)
        self.input_layernorm = RMSNorm(hidden_size, w_input_layernorm, eps=rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(hidden_size, w_post_attention_layernorm, eps=rms_norm_eps)
        self.self_attn = Qwen3MultiHeadAttention(
            hidden_size, num_attention_heads, num_kv_heads, head_dim, wq, wk, wv, wo, q_norm, k_norm,
            max_seq_len=max_seq_len, theta=theta, rms_norm_eps=rms_norm_eps)

    def __call__(self, x: torch.Tensor, mask: torch.Tensor | str | None = None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), mask)
        return h + self.mlp(self.post_attention_layernorm(h))


class Qwen3ModelWeek1:
    def __init__(self, mlx_model: Any):
        args = mlx_model.args
        self.num_hidden_layers = args.num_hidden_layers
        self.hidden_size = args.hidden_size
        self.vocab_size = args.vocab_size
        self.precision = torch.bfloat16
        self.embedding = Embedding(self.vocab_size, self.hidden_size, dequantize_linear(mlx_model.model.embed_tokens))
        self.layers_inner = []
        for layer in mlx_model.model.layers:
            attn, mlp = layer.self_attn, layer.mlp
            self.layers_inner.append(Qwen3TransformerBlock(
                num_attention_heads=args.num_attention_heads, num_kv_heads=args.num_key_value_heads,
                hidden_size=args.hidden_size, head_dim=args.head_dim, intermediate_size=args.intermediate_size,
                rms_norm_eps=args.rms_norm_eps,
                wq=dequantize_linear(attn.q_proj), wk=dequantize_linear(attn.k_proj),
                wv=dequantize_linear(attn.v_proj), wo=dequantize_linear(attn.o_proj),
                q_norm=attn.q_norm.weight, k_norm=attn.k_norm.weight,
                w_gate=dequantize_linear(mlp.gate_proj), w_up=dequantize_linear(mlp.up_proj),
                w_down=dequantize_linear(mlp.down_proj),
                w_input_layernorm=layer.input_layernorm.weight,
                w_post_attention_layernorm=layer.post_attention_layernorm.weight,
                max_seq_len=args.max_position_embeddings, theta=args.rope_theta))
        self.norm = RMSNorm(args.hidden_size, mlx_model.model.norm.weight, eps=args.rms_norm_eps)
        self.w_lm_head = None if args.tie_word_embeddings else dequantize_linear(mlx_model.lm_head)
        self.mlx_model = mlx_model

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        h = self.embedding(inputs)
        for block in self.layers_inner:
            h = block(h, mask="causal")
        h = self.norm(h)
        return linear(h, self.w_lm_head) if self.w_lm_head is not None else self.embedding.as_linear(h)
