"""Mixture-of-experts MLP block for Qwen3-MoE (reference: src/tiny_llm_ref/moe.py:7-89, tests_refsol/test_week_3_day_6.py).

``w_experts`` are QuantizedWeights whose tensors carry a leading expert dimension: weight [E, out, in/8],
scales / biases [E, out, in/128].  The grouped product is ONE launch of the W4 GEMV with an expert index per activation
row (tl_gather_quantized_matvec), so no sorting / un-sorting of the rows is needed (the reference sorts them because
mx.gather_qmm wants grouped right-hand sides).
"""

import torch

from ._ext import tiny_llm_ext_hip
from .basics import silu
from .quantize import QuantizedWeights, quantized_linear

__all__ = ["grouped_expert_linear", "route_topk", "Moe"]


def grouped_expert_linear(x: torch.Tensor, w_experts: QuantizedWeights, expert_ids: torch.Tensor) -> torch.Tensor:
    """out[..., :] = x[..., :] @ dequant(w_experts[expert_ids[...]]).T  (one expert per row of x)."""
    *lead, d = x.shape
    flat = x.reshape(-1, d).contiguous()
    ids = expert_ids.reshape(-1).to(torch.int32).contiguous()
    if ids.numel() != flat.shape[0]:
        raise ValueError("expert_ids must hold one expert per activation row")
    out = tiny_llm_ext_hip.gather_quantized_matvec(w_experts.scales, w_experts.biases, w_experts.group_size,
                                                   w_experts.bits, flat, w_experts.weight, ids)
    return out.reshape(*lead, out.shape[-1])


def route_topk(x: torch.Tensor, w_router: QuantizedWeights, top_k: int, norm_topk_prob: bool = False):
    """Router: softmax over all experts in fp32, the top_k experts of every token and their probabilities (renormalised over
    the selected experts when norm_topk_prob).  Returns (probs [..., E], expert_ids [..., top_k], scores [..., top_k])."""
    logits = quantized_linear(x, w_router)
    probs = torch.softmax(logits.to(torch.float32), dim=-1).to(logits.dtype)
    scores, expert_ids = torch.topk(probs, top_k, dim=-1)
    if norm_topk_prob:
        scores = scores / scores.sum(dim=-1, keepdim=True)
    return probs, expert_ids, scores


class Moe:
    def __init__(self, w_router: QuantizedWeights, w_gate: QuantizedWeights, w_up: QuantizedWeights,
                 w_down: QuantizedWeights, num_experts_per_tok: int, norm_topk_prob: bool = False):
        self.w_router, self.w_gate, self.w_up, self.w_down = w_router, w_gate, w_up, w_down
        self.num_experts_per_tok = num_experts_per_tok
        self.norm_topk_prob = norm_topk_prob

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        B, L, D = x.shape
        k = self.num_experts_per_tok
        _, expert_ids, scores = route_topk(x, self.w_router, k, self.norm_topk_prob)
        rows = x[:, :, None, :].expand(B, L, k, D).reshape(-1, D)
        ids = expert_ids.reshape(-1)
        gate = grouped_expert_linear(rows, self.w_gate, ids)
        up = grouped_expert_linear(rows, self.w_up, ids)
        y = grouped_expert_linear(silu(gate) * up, self.w_down, ids).reshape(B, L, k, D)
        return (y * scores[..., None]).sum(dim=-2)
