"""Week-3 Qwen3: paged KV + paged attention (reference: src/tiny_llm_ref/qwen3_week3.py)."""

from typing import Any

import torch

from .attention import paged_attention
from .embedding import QuantizedEmbedding
from .kv_cache import TinyKvCache
from .moe import Moe
from .paged_kv_cache import TinyKvPagedCache, TinyKvPagedPool
from .quantize import QuantizedWeights, quantized_linear
from .week2_kernels import FastRMSNorm, FastRoPE, decode_attention_custom, scaled_dot_product_attention, swiglu


class Qwen3MultiHeadAttention:
    def __init__(self, hidden_size, num_heads, num_kv_heads, head_dim, wq, wk, wv, wo, q_norm, k_norm,
                 max_seq_len: int = 32768, theta: int = 1000000, rms_norm_eps: float = 1e-5,
                 use_paged_attention: bool = True):
        assert num_heads % num_kv_heads == 0, (
            f"num_heads {num_heads} must be divisible by num_kv_heads {num_kv_heads}")
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.scale = head_dim ** -0.5
        self.wq, self.wk, self.wv, self.wo = wq, wk, wv, wo
        self.rope = FastRoPE(head_dim, max_seq_len, theta)
        self.q_norm = FastRMSNorm(head_dim, q_norm, eps=rms_norm_eps)
        self.k_norm = FastRMSNorm(head_dim, k_norm, eps=rms_norm_eps)
        self.use_paged_attention = use_paged_attention

    def __call__(self, x, offsets, cache: TinyKvCache, mask=None) -> torch.Tensor:
        B, L, _ = x.shape
        q = self.q_norm(quantized_linear(x, self.wq).reshape(B, L, self.num_heads, self.head_dim))
        k = self.k_norm(quantized_linear(x, self.wk).reshape(B, L, self.num_kv_heads, self.head_dim))
        v = quantized_linear(x, self.wv).reshape(B, L, self.num_kv_heads, self.head_dim)
        q = self.rope(q, offset=offsets).transpose(1, 2)
        k = self.rope(k, offset=offsets).transpose(1, 2)
        v = v.transpose(1, 2)
        if self.use_paged_attention:
            meta = cache.update_and_fetch_paged(k, v, mask_length=L, mask=mask)
            mixed = paged_attention(
                q, meta.key_pages, meta.value_pages, meta.block_table, meta.context_lens, meta.page_size,
                scale=self.scale, mask=meta.mask,
                host_block_rows=getattr(meta, "host_block_rows", None),
                host_context_lens=getattr(meta, "host_context_lens", None))
        else:
            key, value, _, mask = cache.update_and_fetch(k, v, mask_length=L, mask=mask)
            if L <= 8 and key.shape[-2] <= 256:
                mixed = decode_attention_custom(q, key, value, scale=self.scale, mask=mask)
            else:
                mixed = scaled_dot_product_attention(q, key, value, scale=self.scale, mask=mask)
        return quantized_linear(mixed.transpose(1, 2).reshape(B, L, self.num_heads * self.head_dim), self.wo)


class Qwen3MLP:
    def __init__(self, dim, hidden_dim, w_gate: QuantizedWeights, w_up: QuantizedWeights, w_down: QuantizedWeights):
        self.dim = dim
        self.hidden_dim = hidden_dim
        self.w_gate, self.w_up, self.w_down = w_gate, w_up, w_down

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return quantized_linear(swiglu(quantized_linear(x, self.w_gate), quantized_linear(x, self.w_up)), self.w_down)


class Qwen3TransformerBlock:
    def __init__(self, num_attention_heads, num_kv_heads, hidden_size, head_dim, rms_norm_eps, wq, wk, wv, wo,
                 q_norm, k_norm, w_input_layernorm, w_post_attention_layernorm, mlp, max_seq_len: int = 32768,
                 theta: int = 1000000, use_paged_attention: bool = True):
        self.num_attention_heads = num_attention_heads
        self.hidden_size = hidden_size
        self.mlp = mlp
        self.input_layernorm = FastRMSNorm(hidden_size, w_input_layernorm, eps=rms_norm_eps)
        self.post_attention_layernorm = FastRMSNorm(hidden_size, w_post_attention_layernorm, eps=rms_norm_eps)
        self.self_attn = Qwen3MultiHeadAttention(
            hidden_size, num_attention_heads, num_kv_heads, head_dim, wq, wk, wv, wo, q_norm, k_norm,
            max_seq_len=max_seq_len, theta=theta, rms_norm_eps=rms_norm_eps,
            use_paged_attention=use_paged_attention)

    def __call__(self, x, offset, cache: TinyKvCache, mask=None) -> torch.Tensor:
        h = x + self.self_attn(self.input_layernorm(x), offset, cache, mask)
        return h + self.mlp(self.post_attention_layernorm(h))


def is_qwen3_moe_sparse_layer(args: Any, layer_idx: int) -> bool:
    return (
        getattr(args, "num_experts", 0) > 0
        and layer_idx not in getattr(args, "mlp_only_layers", [])
        and (layer_idx + 1) % getattr(args, "decoder_sparse_step", 1) == 0
    )


class Qwen3ModelWeek3:
    """Every layer owns one page pool; a request gets one ``TinyKvPagedCache`` per layer on it
    (reference qwen3_week3.py:218-338).  Sparse (Qwen3-MoE) layers use the grouped-expert block of moe.py."""

    def __init__(self, mlx_model: Any, page_size: int = 128, enable_paged_attention: bool = True):
        args = mlx_model.args
        self.num_hidden_layers = args.num_hidden_layers
        self.hidden_size = args.hidden_size
        self.vocab_size = args.vocab_size
        self.page_size = page_size
        self.page_pools = [TinyKvPagedPool(page_size=page_size) for _ in range(self.num_hidden_layers)]
        self.precision = torch.bfloat16

        def w4(layer: Any) -> QuantizedWeights:
            return QuantizedWeights.from_mlx_layer(layer, use_simdgroup_matmul=True, use_split_k_matmul=True)

        self.embedding = QuantizedEmbedding(
            self.vocab_size, self.hidden_size, w4(mlx_model.model.embed_tokens), use_custom_kernel=True)
        self.layers_inner = []
        for index, layer in enumerate(mlx_model.model.layers):
            attn, mlp = layer.self_attn, layer.mlp
            if is_qwen3_moe_sparse_layer(args, index):
                # Qwen3-MoE: router + grouped-expert SwiGLU (reference qwen3_week3.py:258-272, moe.py:39-89)
                mlp_block = Moe(w_router=w4(mlp.gate), w_gate=w4(mlp.switch_mlp.gate_proj),
                                w_up=w4(mlp.switch_mlp.up_proj), w_down=w4(mlp.switch_mlp.down_proj),
                                num_experts_per_tok=args.num_experts_per_tok,
                                norm_topk_prob=getattr(args, "norm_topk_prob", False))
            else:
                mlp_block = Qwen3MLP(args.hidden_size, args.intermediate_size, w4(mlp.gate_proj), w4(mlp.up_proj),
                                     w4(mlp.down_proj))
            self.layers_inner.append(Qwen3TransformerBlock(
                num_attention_heads=args.num_attention_heads, num_kv_heads=args.num_key_value_heads,
                hidden_size=args.hidden_size, head_dim=args.head_dim, rms_norm_eps=args.rms_norm_eps,
                wq=w4(attn.q_proj), wk=w4(attn.k_proj), wv=w4(attn.v_proj), wo=w4(attn.o_proj),
                q_norm=attn.q_norm.weight, k_norm=attn.k_norm.weight,
                w_input_layernorm=layer.input_layernorm.weight,
                w_post_attention_layernorm=layer.post_attention_layernorm.weight,
                mlp=mlp_block,
                max_seq_len=args.max_position_embeddings, theta=args.rope_theta,
                use_paged_attention=enable_paged_attention))
        self.norm = FastRMSNorm(args.hidden_size, weight=mlx_model.model.norm.weight, eps=args.rms_norm_eps)
        self.w_lm_head = None if args.tie_word_embeddings else w4(mlx_model.lm_head)
        self.mlx_model = mlx_model

    def create_kv_cache(self) -> list[TinyKvCache]:
        return [TinyKvPagedCache(pool=pool) for pool in self.page_pools]

    def __call__(self, inputs: torch.Tensor, offset, cache: list[TinyKvCache],
                 logits_to_keep: int | None = None) -> torch.Tensor:
        h = self.embedding(inputs)
        for block, layer_cache in zip(self.layers_inner, cache):
            h = block(h, offset, layer_cache, mask="causal")
        if logits_to_keep is not None:
            if logits_to_keep <= 0:
                raise ValueError("logits_to_keep must be positive")
            h = h[:, -logits_to_keep:, :]
        h = self.norm(h)
        return quantized_linear(h, self.w_lm_head) if self.w_lm_head is not None else self.embedding.as_linear(h)
