"""torch-CPU restatement of the ONE path of the reference that runs on a CPU at the Qwen3-4B shape: `Qwen3ModelWeek2` at its
`kv-cache` checkpoint -- dense (dequantised) bf16 linears, readable RMSNorm / RoPE / SiLU, attention in fp32, concatenating KV cache
(reference: src/tiny_llm_ref/qwen3_week2.py:96-146 attention block, :357-392 model loop; benches/bench.py:158-169 restricts the CPU
device to exactly this checkpoint; SURVEY.md section 8d names it as the baseline beside the GPU number).

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY: `bench.py`'s `cpu_baseline.torch_week2_kv_cache` leg times it on the GPU box's host
cores.  It is a restatement on torch (MLX cannot be installed here), not the reference's MLX CPU stream, and is labelled so.  The
product never imports this file.

Per-op definitions followed (file:line in /root/reference/src/tiny_llm_ref):
  * linear            basics.py:10-19           x @ W^T in the activations' dtype (bf16 storage, the backend accumulates wider)
  * RMSNorm           layer_norm.py:10-15       fp32 x * rsqrt(mean x^2 + eps), cast to bf16, THEN * weight in bf16 (two roundings)
  * RoPE              positional_encoding.py:4-66  non-traditional pairs (i, i + D/2), fp32 cos / sin tables, result cast to bf16
  * SiLU / MLP        basics.py:21-26, qwen3_week2.py:150-177   silu(gate) * up in bf16, then down
  * attention         qwen3_week2.py:138-144 + attention.py:24-66   q / k / v upcast to fp32, GQA by head grouping, causal mask aligned to the
                      END of the context for L > 1 (none for L == 1, qwen3_week2.py:373), softmax fp32, result cast to bf16
  * KV cache          kv_cache.py (TinyKvFullCache): concatenate along the sequence axis
  * dequantisation    quantize.py:103-121       fp32 q * scale + bias, one cast to bf16 (done once, by the caller)
"""

from __future__ import annotations

import time

import torch


class TorchWeek2KvCacheCPU:
    """dense: {"embed": [V, H] bf16, "norm": [H], "layers": [{"q","k","v","o","gate","up","down": [out, in] bf16,
    "q_norm","k_norm": [D], "input_norm","post_norm": [H]}], optional "lm_head": [V, H]} -- CPU tensors."""

    def __init__(self, cfg: dict, dense: dict, linear_dtype: torch.dtype = torch.bfloat16):
        """linear_dtype: storage of the linears' operands.  bf16 is the reference's; float32 (same bf16-VALUED weights, each product
        rounded back to bf16) is for hosts whose torch build has no fast bf16 GEMM -- the arithmetic stays that of the bf16 path up
        to the accumulation width, which the reference's backend does not pin either."""
        self.cfg, self.w = cfg, dense
        self.ld = linear_dtype
        self.k_cache = [None] * cfg["num_hidden_layers"]
        self.v_cache = [None] * cfg["num_hidden_layers"]
        self.offset = 0
        D, half = cfg["head_dim"], cfg["head_dim"] // 2
        self.half = half
        self.inv_freq = torch.pow(torch.tensor(float(cfg["rope_theta"])), -torch.arange(0, half, dtype=torch.float32) / half)

    @staticmethod
    def _rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
        f = x.float()
        f = f * torch.rsqrt(f.pow(2).mean(dim=-1, keepdim=True) + eps)
        return f.to(x.dtype) * w.to(x.dtype)

    def _rope(self, x: torch.Tensor, start: int) -> torch.Tensor:  # x [L, H, D]
        L = x.shape[0]
        ang = torch.outer(torch.arange(start, start + L, dtype=torch.float32), self.inv_freq)  # [L, D/2]
        cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
        x1, x2 = x[..., :self.half].float(), x[..., self.half:].float()
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1).to(x.dtype)

    def _linear(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        return (x.to(self.ld) @ w.T).to(torch.bfloat16)

    @torch.no_grad()
    def forward(self, tokens) -> torch.Tensor:
        """Feed `tokens` (appended to the cache); returns the LAST row's logits [V] (bf16), like --prefill-logits last."""
        cfg = self.cfg
        Hq, Hkv, D, eps = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"], float(cfg["rms_norm_eps"])
        rep = Hq // Hkv
        ids = torch.as_tensor(list(tokens), dtype=torch.long)
        L = ids.numel()
        h = self.w["embed"][ids]  # [L, hidden] bf16
        for i, lw in enumerate(self.w["layers"]):
            x = self._rms_norm(h, lw["input_norm"], eps)
            q = self._linear(x, lw["q"]).reshape(L, Hq, D)
            k = self._linear(x, lw["k"]).reshape(L, Hkv, D)
            v = self._linear(x, lw["v"]).reshape(L, Hkv, D)
            q = self._rope(self._rms_norm(q, lw["q_norm"], eps), self.offset)
            k = self._rope(self._rms_norm(k, lw["k_norm"], eps), self.offset)
            self.k_cache[i] = k if self.k_cache[i] is None else torch.cat([self.k_cache[i], k], dim=0)
            self.v_cache[i] = v if self.v_cache[i] is None else torch.cat([self.v_cache[i], v], dim=0)
            S = self.k_cache[i].shape[0]
            qf = q.float().reshape(L, Hkv, rep, D).permute(1, 2, 0, 3)          # [Hkv, rep, L, D]
            kf = self.k_cache[i].float().permute(1, 0, 2)                       # [Hkv, S, D]
            vf = self.v_cache[i].float().permute(1, 0, 2)
            scores = torch.einsum("grld,gsd->grls", qf, kf) * (D ** -0.5)
            if L > 1:  # causal, aligned to the end of the context (attention.py:24-27: k = S - L)
                keep = torch.arange(S)[None, :] <= (S - L + torch.arange(L))[:, None]
                scores = scores.masked_fill(~keep, float("-inf"))
            att = torch.einsum("grls,gsd->grld", torch.softmax(scores, dim=-1), vf)
            att = att.permute(2, 0, 1, 3).reshape(L, Hq * D).to(h.dtype)
            h = h + self._linear(att, lw["o"])
            x = self._rms_norm(h, lw["post_norm"], eps)
            g, u = self._linear(x, lw["gate"]), self._linear(x, lw["up"])
            h = h + self._linear((g * torch.sigmoid(g.float()).to(g.dtype)) * u, lw["down"])
        self.offset += L
        last = self._rms_norm(h[-1:], self.w["norm"], eps)
        head = self.w.get("lm_head")
        if head is None:
            head = self.w.get("embed_linear", self.w["embed"])  # tied head; "embed_linear": the table in linear_dtype
        return self._linear(last, head)[0]

    def timed_decode(self, first_token: int, steps: int, fed=None):
        """`steps` greedy decode steps (or teacher-forced on `fed`); returns (seconds, ids, logits of the first step)."""
        ids, first_logits = [int(first_token)], None
        t0 = time.perf_counter()
        for s in range(steps):
            logits = self.forward([ids[-1] if fed is None else int(fed[s])])
            if first_logits is None:
                first_logits = logits
            ids.append(int(torch.argmax(logits.float())))
        return time.perf_counter() - t0, ids[1:], first_logits
