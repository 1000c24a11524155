#!/usr/bin/env python3
"""Operator microbench of decode attention over a paged KV cache: batch 1, 32 query / 8 KV heads, one query token,
head dim 128, contexts 128 .. 32,768 (reference: benches/bench_week3_attention.py:25-133 -- shape constants, `build_case`,
inputs `normal` with seed = seed + context, the max-abs-error check of the paged result against the dense one).

Variants (the reference compares dense-gather / direct-paged / mlx-fused):
  dense_gather  gather the pages into dense K/V, then the readable grouped attention in torch   (what Week 3 replaces)
  direct_paged  the public operator `paged_attention`                                         (tl_paged_attention, L = 1)
  engine_fused  the engine's attention launch: q/k-norm + RoPE + KV append + attention (+ merge) (tl_decode_attention_fused)
Reported per context: median us per call, K/V bytes / time (algorithmic: ctx x 4,096 B per layer) as GB/s and
as a fraction of 8 TB/s, and the max abs error of the operator against the dense result (reference's published numbers on
M4 Pro: 0.0044 at 128, 0.00195 at 1,024; benchmark_results/m4-pro-qwen3-4b-week3-attention-mlx-0.32.0.json).

Timing as in bench_week2_operators.py: `iterations` calls captured into one graph, replayed between HIP events, over
rotating copies of the page pools (> 512 MB in total where the context allows).
"""

from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path
from statistics import median

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

QUERY_HEADS, KV_HEADS, HEAD_DIM = 32, 8, 128
HBM_PEAK_GBPS = 8000.0


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, nargs="+", default=[128, 1024, 8192, 32768])
    ap.add_argument("--page-size", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--json-output", type=Path)
    ap.add_argument("--mfma-rows", action="store_true",
                    help="lab column (not in the reference's table): the same decode through the MFMA FlashAttention kernel of the "
                         "public operator -- every head's single query row padded to 16 rows, non-causal -- i.e. what the prefill "
                         "kernel's K/V pipeline sustains when the arithmetic is negligible; the yardstick for an MFMA decode kernel")
    args = ap.parse_args(argv)
    if args.warmup < 0 or args.iterations <= 0 or args.repeats <= 0:
        ap.error("--warmup must be non-negative; --iterations and --repeats must be positive")
    return args


def build_case(context: int, page_size: int, seed: int, copies: int):
    """`copies` independent page pools holding the same logical K/V through scattered physical pages."""
    import torch

    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed + context)
    bf = torch.bfloat16
    q = torch.randn((1, QUERY_HEADS, 1, HEAD_DIM), generator=gen, device="cuda").to(bf)
    keys = torch.randn((1, KV_HEADS, context, HEAD_DIM), generator=gen, device="cuda").to(bf)
    values = torch.randn((1, KV_HEADS, context, HEAD_DIM), generator=gen, device="cuda").to(bf)
    pages = (context + page_size - 1) // page_size
    pools = []
    for c in range(copies):
        perm = torch.randperm(pages + 2, generator=gen, device="cuda")[:pages].to(torch.int32)
        kp = torch.zeros((pages + 2, KV_HEADS, page_size, HEAD_DIM), dtype=bf, device="cuda")
        vp = torch.zeros_like(kp)
        pad = pages * page_size - context
        kpad = torch.nn.functional.pad(keys[0], (0, 0, 0, pad)).reshape(KV_HEADS, pages, page_size, HEAD_DIM).transpose(0, 1)
        vpad = torch.nn.functional.pad(values[0], (0, 0, 0, pad)).reshape(KV_HEADS, pages, page_size, HEAD_DIM).transpose(0, 1)
        kp[perm.long()] = kpad
        vp[perm.long()] = vpad
        table = torch.full((1, pages + 1), -1, dtype=torch.int32, device="cuda")
        table[0, :pages] = perm
        pools.append((kp, vp, table))
    return q, keys, values, pools


def dense_attention(q, keys, values):
    """Readable grouped attention (reference attention.py:30-66) in fp32 on the dense tensors."""
    import torch

    rep = QUERY_HEADS // KV_HEADS
    qf = q.float().reshape(1, KV_HEADS, rep, 1, HEAD_DIM)
    scores = torch.matmul(qf, keys.float().unsqueeze(2).transpose(-1, -2)) * HEAD_DIM ** -0.5
    return torch.matmul(torch.softmax(scores, dim=-1), values.float().unsqueeze(2)).reshape(1, QUERY_HEADS, 1, HEAD_DIM).to(q.dtype)


def timed(call, warmup, iterations, repeats):
    import torch

    for i in range(warmup):
        call(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(iterations):
                call(i)
    torch.cuda.synchronize()
    samples = []
    for _ in range(repeats):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        samples.append(a.elapsed_time(b) * 1e3 / iterations)
    return median(samples), samples


def main(argv=None) -> dict:
    args = parse_args(argv)
    import torch

    import tiny_llm_ext_hip as ext

    if not torch.cuda.is_available():
        raise SystemExit("bench_week3_attention needs a GPU: the extension has no CPU fallback")
    ext.load_library(str(ROOT))
    rows = []
    print("| Context | dense gather us | direct paged us | engine fused us | engine KV GB/s (frac of 8 TB/s) | paged max abs err |")
    print("|---:|---:|---:|---:|---:|---:|")
    for context in args.contexts:
        kv_bytes = 2 * context * KV_HEADS * HEAD_DIM * 2
        copies = max(1, min(16, (512 << 20) // max(kv_bytes, 1)))
        q, keys, values, pools = build_case(context, args.page_size, args.seed, copies)
        ctx = torch.tensor([context], dtype=torch.int32, device="cuda")
        expected = dense_attention(q, keys, values)
        qn = q.reshape(QUERY_HEADS, 1, HEAD_DIM).contiguous()

        def direct(i):
            kp, vp, table = pools[i % copies]
            return ext.paged_attention(qn, kp, vp, table, ctx, HEAD_DIM ** -0.5, True, num_kv_heads=KV_HEADS,
                                       num_heads=QUERY_HEADS, max_context_hint=context)

        got = direct(0).reshape(1, QUERY_HEADS, 1, HEAD_DIM)
        err = float((got.float() - expected.float()).abs().max())
        if not torch.allclose(got.float(), expected.float(), rtol=2e-2, atol=2e-2):  # the reference's own acceptance check
            raise AssertionError("direct paged attention does not match dense attention")

        def gather(i):
            kp, vp, table = pools[i % copies]
            ids = table[0, :-1].long()
            k = kp[ids].transpose(0, 1).reshape(1, KV_HEADS, -1, HEAD_DIM)[:, :, :context]
            v = vp[ids].transpose(0, 1).reshape(1, KV_HEADS, -1, HEAD_DIM)[:, :, :context]
            return dense_attention(q, k, v)

        # the engine's launch appends one token: context - 1 cached tokens + the new one = the same `context` attended tokens
        qkv = torch.randn((1, (QUERY_HEADS + 2 * KV_HEADS) * HEAD_DIM), device="cuda").to(torch.bfloat16)
        ones = torch.ones((HEAD_DIM,), dtype=torch.bfloat16, device="cuda")
        cached = torch.tensor([context - 1], dtype=torch.int32, device="cuda")

        def fused(i):
            kp, vp, table = pools[i % copies]
            return ext.decode_attention_fused(qkv, ones, ones, kp, vp, table, cached, num_heads=QUERY_HEADS, num_kv_heads=KV_HEADS,
                                              rope_theta=1e6, eps=1e-6, max_context=context - 1)

        _, info = fused(0)
        med = {}
        samples = {}
        variants = [("dense_gather", gather), ("direct_paged", direct), ("engine_fused", fused)]
        if args.mfma_rows:
            q16 = torch.zeros((QUERY_HEADS, 16, HEAD_DIM), dtype=torch.bfloat16, device="cuda")
            q16[:, 0] = qn[:, 0]

            def mfma_rows(i):
                kp, vp, table = pools[i % copies]
                return ext.paged_attention(q16, kp, vp, table, ctx, HEAD_DIM ** -0.5, False, num_kv_heads=KV_HEADS,
                                           num_heads=QUERY_HEADS, max_context_hint=context)

            got16 = mfma_rows(0)[:, 0].reshape(1, QUERY_HEADS, 1, HEAD_DIM)
            if not torch.allclose(got16.float(), expected.float(), rtol=2e-2, atol=2e-2):
                raise AssertionError("the 16-row MFMA call does not match dense attention on its real row")
            variants.append(("mfma_rows", mfma_rows))
        for name, call in variants:
            med[name], samples[name] = timed(call, args.warmup, args.iterations, args.repeats)
        if args.mfma_rows:
            print(f"|   lab: MFMA FlashAttention kernel, 16-row queries: {med['mfma_rows']:.2f} us = "
                  f"{kv_bytes / med['mfma_rows'] / 1e3:.0f} GB/s of K/V |", flush=True)
        gbps = kv_bytes / med["engine_fused"] / 1e3
        print(f"| {context} | {med['dense_gather']:.2f} | {med['direct_paged']:.2f} | {med['engine_fused']:.2f} | "
              f"{gbps:.0f} ({gbps / HBM_PEAK_GBPS:.3f}) | {err:.5f} |", flush=True)
        rows.append({"context": context, "kv_bytes": kv_bytes, "pool_copies": copies, "medians_us": med, "samples_us": samples,
                     "engine_GBps": gbps, "engine_frac_of_8TBps": gbps / HBM_PEAK_GBPS,
                     "direct_paged_GBps": kv_bytes / med["direct_paged"] / 1e3, "direct_max_abs_error": err, "engine_plan": info})
        del pools
        torch.cuda.empty_cache()
    out = {"shape": {"batch": 1, "query_heads": QUERY_HEADS, "kv_heads": KV_HEADS, "query_len": 1, "head_dim": HEAD_DIM,
                     "page_size": args.page_size}, "warmup": args.warmup, "iterations": args.iterations, "repeats": args.repeats,
           "results": rows}
    if args.json_output:
        args.json_output.parent.mkdir(parents=True, exist_ok=True)
        args.json_output.write_text(json.dumps(out, indent=1))
        print(f"Wrote {args.json_output}")
    return out


if __name__ == "__main__":
    main()
