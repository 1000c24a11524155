"""Affine W4 quantiser and synthetic Qwen3-shaped checkpoints.

No real checkpoint can be downloaded where this runs, so benches and tests build random-weight models with
the exact tensor layout ``mlx_lm.load`` would hand the reference models (the attribute tree read by
qwen3_week2.py:288-350): ``model.args.*`` plus, per linear, ``weight`` (packed uint32 bits held in an
int32 tensor), ``scales``, ``biases``, ``group_size``, ``bits``.
"""

from __future__ import annotations

from types import SimpleNamespace

import math

import torch

QWEN3_CONFIGS = {
    # public Qwen3 configs (values pinned for 4B in the reference's benchmark JSON)
    "qwen3-4b": dict(hidden_size=2560, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8,
                     head_dim=128, intermediate_size=9728, vocab_size=151936, rope_theta=1000000,
                     rms_norm_eps=1e-6, max_position_embeddings=40960, tie_word_embeddings=True),
    "qwen3-0.6b": dict(hidden_size=1024, num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8,
                       head_dim=128, intermediate_size=3072, vocab_size=151936, rope_theta=1000000,
                       rms_norm_eps=1e-6, max_position_embeddings=40960, tie_word_embeddings=True),
}


def quantize(w: torch.Tensor, group_size: int = 128, bits: int = 4):
    """Group-wise affine quantisation in the MLX packing: (packed [K, N/8] int32 bits, scales, biases).

    Restates ``mx.quantize`` (mlx 0.32): per group, scale = (max-min)/15 signed towards the larger-magnitude
    edge, that edge becomes the bias after snapping it to a multiple of the scale; codes are packed eight
    per 32-bit word with element 8j+i in bits [4i, 4i+4) (reference quantize.py:113-115)."""
    if bits != 4:
        raise ValueError("only 4-bit quantisation is supported")
    *lead, N = w.shape  # leading dimensions (rows; an expert stack [E, K, N]) are quantised independently
    if N % group_size:
        raise ValueError("last dimension must be divisible by group_size")
    dtype = w.dtype
    g = w.to(torch.float32).reshape(*lead, N // group_size, group_size)
    hi = g.amax(dim=-1)
    lo = g.amin(dim=-1)
    scale = torch.clamp((hi - lo) / 15.0, min=1e-7)
    lower_side = lo.abs() > hi.abs()
    scale = torch.where(lower_side, scale, -scale)
    edge = torch.where(lower_side, lo, hi)
    q0 = torch.round(edge / scale)
    snap = q0 != 0
    scale = torch.where(snap, edge / torch.where(snap, q0, torch.ones_like(q0)), scale)
    bias = torch.where(snap, edge, torch.zeros_like(edge))
    scale = scale.to(dtype)
    bias = bias.to(dtype)
    s32 = scale.to(torch.float32)
    safe = torch.where(s32 == 0, torch.ones_like(s32), s32)
    codes = torch.clamp(torch.round((g - bias.to(torch.float32)[..., None]) / safe[..., None]), 0, 15)
    codes = codes.to(torch.int64).reshape(*lead, N // 8, 8)
    shifts = torch.arange(0, 32, 4, dtype=torch.int64, device=w.device)
    words = (codes << shifts).sum(dim=-1)  # < 2^32
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    return words, scale, bias


def _qlayer(weight: torch.Tensor) -> SimpleNamespace:
    packed, scales, biases = quantize(weight)
    return SimpleNamespace(weight=packed, scales=scales, biases=biases, group_size=128, bits=4)


def synthetic_qwen3(config: dict | str, seed: int = 0, sigma: float = 0.02, device: str = "cuda",
                    norm_jitter: float = 0.05, embed_sigma: float | None = None, residual_gain: float = 1.0,
                    head_permutation: tuple[int, int] | None = None) -> SimpleNamespace:
    """Random-weight Qwen3 in the mlx_lm object shape.  w ~ N(0, sigma) in bf16 -> W4 g128.

    embed_sigma / residual_gain build a PEAKED checkpoint: N(0, sigma) everywhere gives flat logits (the layers' outputs swamp the
    embedding, the top two logits of 151,936 lie within a rounding error of each other, and "the greedy id equals the truth's" is
    a coin toss there).  With a larger embedding (embed_sigma) and the two projections that write the residual stream (o_proj,
    down_proj) scaled by residual_gain < 1, the stream keeps a clear component along the input token's embedding row and the tied
    head answers with a margin far above any rounding error: greedy ids can be REQUIRED to equal the float64 truth's.

    head_permutation = (a, b), a coprime to the vocabulary: an UNTIED head whose row (a t + b) mod V is the embedding row of token t
    (the quantised tensors, moved row-wise: exact).  A tied head on such a checkpoint echoes its input token for ever (round 3's
    checker produced nine copies of one id with a margin of 460 logit units -- it could not see a kernel that is wrong by a hundred);
    with the permuted head the stream's component along embedding row t votes for token a t + b: the greedy sequence walks the
    permutation, every step answers with a different id, and residual_gain sets how far the top-2 margin stands above the rounding
    error (bench.py / tests/test_engine_qwen4b_gpu.py want 5-50 x the engine's measured error)."""
    cfg = dict(QWEN3_CONFIGS[config]) if isinstance(config, str) else dict(config)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    def linear(out_dim: int, in_dim: int, scale: float = sigma) -> SimpleNamespace:
        w = (torch.randn((out_dim, in_dim), generator=gen, device=device, dtype=torch.float32) * scale)
        return _qlayer(w.to(torch.bfloat16))

    def norm(n: int) -> SimpleNamespace:
        w = 1.0 + norm_jitter * torch.randn((n,), generator=gen, device=device, dtype=torch.float32)
        return SimpleNamespace(weight=w.to(torch.bfloat16))

    hs, inter = cfg["hidden_size"], cfg["intermediate_size"]
    hq, hkv, hd = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    layers = []
    for _ in range(cfg["num_hidden_layers"]):
        layers.append(SimpleNamespace(
            self_attn=SimpleNamespace(
                q_proj=linear(hq * hd, hs), k_proj=linear(hkv * hd, hs), v_proj=linear(hkv * hd, hs),
                o_proj=linear(hs, hq * hd, sigma * residual_gain), q_norm=norm(hd), k_norm=norm(hd)),
            mlp=SimpleNamespace(gate_proj=linear(inter, hs), up_proj=linear(inter, hs), down_proj=linear(hs, inter, sigma * residual_gain)),
            input_layernorm=norm(hs), post_attention_layernorm=norm(hs)))
    model = SimpleNamespace(embed_tokens=linear(cfg["vocab_size"], hs, embed_sigma if embed_sigma is not None else sigma), layers=layers,
                            norm=norm(hs))
    if head_permutation is not None:
        cfg["tie_word_embeddings"] = False
    out = SimpleNamespace(args=SimpleNamespace(**cfg), model=model)
    if head_permutation is not None:
        a, b = head_permutation
        V = cfg["vocab_size"]
        if math.gcd(a, V) != 1:
            raise ValueError("head_permutation: a must be coprime to the vocabulary size")
        target = (torch.arange(V, dtype=torch.int64, device=device) * a + b) % V  # row of the head that holds embedding row t
        e = model.embed_tokens
        head = SimpleNamespace(weight=torch.empty_like(e.weight), scales=torch.empty_like(e.scales), biases=torch.empty_like(e.biases),
                               group_size=128, bits=4)
        head.weight[target], head.scales[target], head.biases[target] = e.weight, e.scales, e.biases
        out.lm_head = head
    elif not cfg.get("tie_word_embeddings", True):
        out.lm_head = linear(cfg["vocab_size"], hs)
    return out
