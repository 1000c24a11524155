"""`mlx_lm.models.qwen3_moe.Qwen3MoeSparseMoeBlock`: the oracle of the reference's optional MoE chapter
(tests_refsol/test_week_3_day_6.py:7-9,95-135).  Router softmax in fp32 (`precise=True`), top-k by partition,
optional renormalisation over the selected experts, grouped SwiGLU experts, probability-weighted sum.
Restated over the facade (fp32 torch); PARITY UNPINNED against real mlx-lm.
"""

from __future__ import annotations

import mlx.core as mx
import mlx.nn as nn

from .switch_layers import SwitchGLU


class Qwen3MoeSparseMoeBlock(nn.Module):
    def __init__(self, args):
        dim = args.hidden_size
        self.num_experts = args.num_experts
        self.top_k = args.num_experts_per_tok
        self.norm_topk_prob = args.norm_topk_prob
        self.gate = nn.Linear(dim, self.num_experts, bias=False)
        self.switch_mlp = SwitchGLU(dim, args.moe_intermediate_size, self.num_experts)

    def __call__(self, x):
        gates = mx.softmax(self.gate(x), axis=-1, precise=True)
        k = self.top_k
        inds = mx.argpartition(-gates, kth=k - 1, axis=-1)[..., :k]
        scores = mx.take_along_axis(gates, inds, axis=-1)
        if self.norm_topk_prob:
            scores = scores / scores.sum(dim=-1, keepdim=True)
        y = self.switch_mlp(x, inds)
        return (y * scores[..., None]).sum(dim=-2)
