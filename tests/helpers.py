"""Test-only helpers shared by the GPU parity tests: move oracle (numpy) checkpoints into the mlx_lm-shaped
object tree the product models and the decode engine consume (attribute names as read by
reference qwen3_week2.py:288-350)."""

from types import SimpleNamespace

import numpy as np
import torch

TINY_CFG = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                intermediate_size=512, vocab_size=1024, rope_theta=1000000, rms_norm_eps=1e-6,
                max_position_embeddings=4096, tie_word_embeddings=True)


def _bf16(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device, torch.bfloat16)


def _qlayer(t, device):
    packed, scales, biases = t
    return SimpleNamespace(weight=torch.from_numpy(np.ascontiguousarray(packed).view(np.int32)).to(device),
                           scales=_bf16(scales, device), biases=_bf16(biases, device), group_size=128, bits=4)


def to_mlx_shaped(cfg: dict, w: dict, device: str = "cuda") -> SimpleNamespace:
    layers = []
    for lw in w["layers"]:
        layers.append(SimpleNamespace(
            self_attn=SimpleNamespace(
                q_proj=_qlayer(lw["q"], device), k_proj=_qlayer(lw["k"], device), v_proj=_qlayer(lw["v"], device),
                o_proj=_qlayer(lw["o"], device), q_norm=SimpleNamespace(weight=_bf16(lw["q_norm"], device)),
                k_norm=SimpleNamespace(weight=_bf16(lw["k_norm"], device))),
            mlp=SimpleNamespace(gate_proj=_qlayer(lw["gate"], device), up_proj=_qlayer(lw["up"], device),
                                down_proj=_qlayer(lw["down"], device)),
            input_layernorm=SimpleNamespace(weight=_bf16(lw["input_norm"], device)),
            post_attention_layernorm=SimpleNamespace(weight=_bf16(lw["post_norm"], device))))
    model = SimpleNamespace(embed_tokens=_qlayer(w["embed"], device), layers=layers,
                            norm=SimpleNamespace(weight=_bf16(w["norm"], device)))
    out = SimpleNamespace(args=SimpleNamespace(**cfg), model=model)
    if "lm_head" in w:
        out.lm_head = _qlayer(w["lm_head"], device)
    return out


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))).astype(np.float32)
