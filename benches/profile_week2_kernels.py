#!/usr/bin/env python3
"""Kernel-group replay of the Week-2 model (reference: benches/profile_week2_kernels.py:24-420 -- same cases, groups,
rotation, report lines and JSON schema): where does the time of the op-by-op model go, per cumulative checkpoint?

A case is CHECKPOINT:PHASE:TOKENS.  For it, each group of kernels is replayed at the model's real shapes and dispatch counts
(every layer's projections / attention / normalisation-position-activation / KV growth in one go), groups in rotated order,
median of synchronised wall-clock samples, shares normalised over the measured groups (reference benchmark_groups,
profile_week2_kernels.py:139-156).  The model object is ``tiny_llm_hip.Qwen3ModelWeek2`` over a synthetic checkpoint of the
named shape (no weights can be downloaded here); the operators are the HIP ones, there is no CPU fallback.

This attributes the time of the REFERENCE-STRUCTURED path (~690 Python-dispatched launches per token).  The fused engine that
bench.py times has its own attribution (`tl_engine_profile_step`, `roofline.per_kind`).
"""

from __future__ import annotations

import argparse
import json
import platform
import sys
from dataclasses import asdict, dataclass
from pathlib import Path
from statistics import median
from time import perf_counter

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

DEFAULT_CASES = ("kv-cache:decode:128", "quantized-matvec:decode:128", "swiglu:decode:128", "decode-attention:decode:128",
                 "decode-attention:prefill:128", "simd-matmul:prefill:128", "simd-matmul:prefill:32", "split-k:prefill:32")
GROUPS = ("projections", "attention", "normalization, position, and activation", "KV growth")


@dataclass(frozen=True)
class ProfileCase:
    checkpoint: str
    phase: str
    tokens: int


@dataclass(frozen=True)
class CategoryResult:
    name: str
    median_us: float
    share: float


def parse_case(value: str) -> ProfileCase:
    try:
        checkpoint, phase, raw = value.split(":")
        tokens = int(raw)
    except ValueError as exc:
        raise argparse.ArgumentTypeError("cases use CHECKPOINT:PHASE:TOKENS") from exc
    if phase not in ("decode", "prefill"):
        raise argparse.ArgumentTypeError("phase must be decode or prefill")
    if tokens <= 0:
        raise argparse.ArgumentTypeError("tokens must be positive")
    return ProfileCase(checkpoint, phase, tokens)


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="Attribute Week 2 time by replaying each real kernel group at its Qwen model "
                                             "shape and dispatch count.")
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--case", action="append", type=parse_case,
                    help="profile CHECKPOINT:PHASE:TOKENS; repeat for more cases (default: the Week 2 bottleneck progression)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--iterations", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json-output", type=Path)
    args = ap.parse_args(argv)
    if args.warmup < 0 or args.iterations <= 0:
        ap.error("warmup cannot be negative and iterations must be positive")
    if args.case is None:
        args.case = [parse_case(v) for v in DEFAULT_CASES]
    return args


def rotations(builders):
    return [builders[k:] + builders[:k] for k in range(len(builders))]


def benchmark_groups(builders, warmup: int, iterations: int, evaluate) -> dict[str, float]:
    """Every group once per round, the order rotated by one each round; median of the timed rounds, in microseconds."""
    orders = rotations(builders)
    for r in range(warmup):
        for _, build in orders[r % len(orders)]:
            evaluate(build())
    samples = {name: [] for name, _ in builders}
    for r in range(iterations):
        for name, build in orders[(warmup + r) % len(orders)]:
            t0 = perf_counter()
            evaluate(build())
            samples[name].append(perf_counter() - t0)
    return {name: median(v) * 1e6 for name, v in samples.items()}


class KernelReplay:
    """Inputs of the model's shapes, and the four kernel groups as lists of outputs (nothing is fed forward between groups:
    each group's cost is measured in isolation, at the dispatch count of one forward pass)."""

    def __init__(self, model, phase: str, tokens: int, seed: int):
        import torch

        from tiny_llm_hip import qwen3_week2 as w2
        from tiny_llm_hip.attention import scaled_dot_product_attention_grouped
        from tiny_llm_hip.basics import linear, silu
        from tiny_llm_hip.quantize import QuantizedWeights, quantized_linear
        from tiny_llm_hip.week2_kernels import decode_attention_custom, swiglu

        self.torch, self.model, self.phase = torch, model, phase
        self.linear, self.silu, self.qlinear, self.qtype = linear, silu, quantized_linear, QuantizedWeights
        self.grouped, self.decode_attention, self.swiglu = scaled_dot_product_attention_grouped, decode_attention_custom, swiglu
        self.max_query, self.max_context = w2.DECODE_ATTENTION_MAX_QUERY, w2.DECODE_ATTENTION_MAX_CONTEXT
        self.rows = 1 if phase == "decode" else tokens
        self.context = tokens
        first = model.layers_inner[0]
        H, Hq, Hkv, D = first.hidden_size, first.self_attn.num_heads, first.self_attn.num_kv_heads, first.self_attn.head_dim
        inter = first.mlp.hidden_dim
        dt = model.precision
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        rnd = lambda *shape: torch.randn(shape, generator=gen, device="cuda").to(dt)
        self.hidden = rnd(1, self.rows, H)
        self.query = rnd(1, Hq, self.rows, D)
        self.key = rnd(1, Hkv, self.context, D)
        self.value = rnd(1, Hkv, self.context, D)
        self.query_rows = self.query.transpose(1, 2).contiguous()
        self.key_rows = rnd(1, self.rows, Hkv, D)
        self.gate = rnd(1, self.rows, inter)
        self.up = rnd(1, self.rows, inter)
        self.tokens = torch.zeros((1, self.rows), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()

    def project(self, x, w):
        return self.qlinear(x, w) if isinstance(w, self.qtype) else self.linear(x, w)

    def projections(self):
        torch, outs, hidden = self.torch, [], self.hidden
        for layer in self.model.layers_inner:
            att = layer.self_attn
            q, k, v = self.project(hidden, att.wq), self.project(hidden, att.wk), self.project(hidden, att.wv)
            att_in = torch.cat((k, v, q[..., k.shape[-1] + v.shape[-1]:]), dim=-1)  # the o projection's input width
            mlp_in = hidden + self.project(att_in, att.wo)
            gate, up = self.project(mlp_in, layer.mlp.w_gate), self.project(mlp_in, layer.mlp.w_up)
            hidden = mlp_in + self.project(gate + up, layer.mlp.w_down)
            outs.extend((k, v))
        last = hidden[:, -1:, :]
        if self.model.w_lm_head is not None:
            outs.append(self.project(last, self.model.w_lm_head))
        else:
            outs.append(self.model.embedding.as_linear(last))
        return outs

    def attention(self):
        torch, outs = self.torch, []
        mask = "causal" if self.phase == "prefill" else None
        for layer in self.model.layers_inner:
            att = layer.self_attn
            custom = att.use_decode_attention and self.rows <= self.max_query and self.context <= self.max_context \
                and not isinstance(mask, torch.Tensor)
            if custom:
                outs.append(self.decode_attention(self.query, self.key, self.value, scale=att.scale, mask=mask))
            else:
                outs.append(self.grouped(self.query.float(), self.key.float(), self.value.float(), scale=att.scale,
                                         mask=mask).to(self.model.precision))
        return outs

    def pointwise(self):
        outs = [self.model.embedding(self.tokens)]
        for layer in self.model.layers_inner:
            att = layer.self_attn
            outs.extend((layer.input_layernorm(self.hidden), layer.post_attention_layernorm(self.hidden),
                         att.q_norm(self.query_rows), att.k_norm(self.key_rows)))
            offset = 0 if att.use_fast_rope else slice(0, self.rows)
            outs.extend((att.rope(self.query_rows, offset=offset), att.rope(self.key_rows, offset=offset)))
            outs.append(self.swiglu(self.gate, self.up) if layer.mlp.use_fast_swiglu else self.silu(self.gate) * self.up)
            outs.extend((self.hidden + self.hidden, self.hidden + self.hidden))
        outs.append(self.model.norm(self.hidden[:, -1:, :]))
        return outs

    def cache(self):
        torch = self.torch
        if self.phase == "prefill":
            return [self.key, self.value]
        pk, pv, nk, nv = self.key[:, :, :-1, :], self.value[:, :, :-1, :], self.key[:, :, -1:, :], self.value[:, :, -1:, :]
        outs = []
        for _ in self.model.layers_inner:
            outs.extend((torch.cat((pk, nk), dim=2), torch.cat((pv, nv), dim=2)))
        return outs


def profile_case(mlx_model, case: ProfileCase, warmup: int, iterations: int, seed: int) -> dict:
    import torch

    from tiny_llm_hip import Qwen3ModelWeek2

    model = Qwen3ModelWeek2(mlx_model, checkpoint=case.checkpoint)
    replay = KernelReplay(model, case.phase, case.tokens, seed)
    builders = [(GROUPS[0], replay.projections), (GROUPS[1], replay.attention), (GROUPS[2], replay.pointwise)]
    if case.phase == "decode":
        builders.append((GROUPS[3], replay.cache))

    def evaluate(outputs):  # the reference's mx.eval(*outputs): everything enqueued has finished
        del outputs
        torch.cuda.synchronize()

    timings = benchmark_groups(builders, warmup, iterations, evaluate)
    total = sum(timings[name] for name, _ in builders)
    categories = [CategoryResult(name, timings[name], timings[name] / total) for name, _ in builders]
    print(f"{case.checkpoint:<18} {case.phase:<7} tokens={case.tokens:<4}")
    for c in categories:
        print(f"  {c.name:<40} {c.median_us:>10.1f} us {c.share:>6.1%}")
    return {"checkpoint": case.checkpoint, "phase": case.phase, "tokens": case.tokens, "attributed_us": total,
            "categories": [asdict(c) for c in categories]}


def main(argv=None) -> dict:
    args = parse_args(argv)
    import torch

    from tiny_llm_hip import qwen3_week2 as w2
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    unknown = [c.checkpoint for c in args.case if c.checkpoint not in w2.WEEK2_CHECKPOINTS]
    if unknown:
        raise SystemExit(f"unknown Week 2 checkpoint: {unknown[0]} (choose from {', '.join(w2.WEEK2_CHECKPOINTS)})")
    if not torch.cuda.is_available():
        raise SystemExit("profile_week2_kernels needs a GPU: the HIP operators have no CPU fallback")
    mlx_model = synthetic_qwen3(dict(QWEN3_CONFIGS[args.model]), seed=args.seed, sigma=0.02, device="cuda")
    print(f"Solution=tiny_llm_hip Model={args.model} (synthetic weights of that shape) torch={torch.__version__}")
    print("Median synchronized kernel-group replay; shares are normalized across the measured groups.")
    profiles = [profile_case(mlx_model, case, args.warmup, args.iterations, args.seed) for case in args.case]
    result = {"schema_version": 1, "solution": "tiny_llm_hip", "model": args.model, "torch_version": torch.__version__,
              "machine": platform.machine(), "platform": platform.platform(), "device": torch.cuda.get_device_name(0),
              "warmup": args.warmup, "iterations": args.iterations, "profiles": profiles}
    if args.json_output is not None:
        args.json_output.parent.mkdir(parents=True, exist_ok=True)
        args.json_output.write_text(json.dumps(result, indent=2) + "\n")
    return result


if __name__ == "__main__":
    main()
