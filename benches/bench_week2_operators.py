#!/usr/bin/env python3
"""Operator microbench of the decode projections: the eight W4A16 GEMV shapes of one Qwen3-4B decode step
(reference: benches/bench_week2_operators.py:336-404 `benchmark_decode_projections` -- q / k / v / o / gate / up / down /
lm head with x ~ N(0,1) bf16 [1, 1, in], 12 warm-up + 60 timed calls, measurement order rotated between variants).

Variants per shape (the reference compares vanilla / optimized / mlx):
  vanilla   one-thread-per-output kernel            (tl_quantized_matmul, use_simdgroup = False)
  operator  the public operator's packed-dot GEMV   (tl_quantized_matmul, rows <= 8: qmv_kernel over the checkpoint layout)
  engine    the decode engine's MFMA GEMV           (tl_decode_linear: qmv3_kernel over the tiled layout -- what bench.py times)
and beside each median the algorithmic HBM rate: 0.53125 B per weight / median, as a fraction of 8 TB/s.

Timing: every variant's `iterations` calls are captured into one graph and replayed between two HIP events, over
ROTATING weight copies (> 512 MB in total, so the 256 MB Infinity Cache never holds the next call's weights); one sample
= elapsed / iterations, `--repeats` samples per variant, median reported.  A sample therefore includes the dependent
kernel boundary (~1.2 us), as a decode step does.  Synthetic weights (N(0, 0.02) bf16 -> affine W4 g128).

    python benches/bench_week2_operators.py [--json-output out.json]
"""

from __future__ import annotations

import argparse
import json
import sys
from itertools import permutations
from pathlib import Path
from statistics import median

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBPS = 8000.0
# (name, out features K, in features N) of Qwen3-4B
PROJECTIONS = (("q projection", 4096, 2560), ("k projection", 1024, 2560), ("v projection", 1024, 2560),
               ("o projection", 2560, 4096), ("gate projection", 9728, 2560), ("up projection", 9728, 2560),
               ("down projection", 2560, 9728), ("lm head", 151936, 2560))


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--iterations", type=int, default=60)
    ap.add_argument("--repeats", type=int, default=6, help="samples per variant; the order of the variants rotates between them")
    ap.add_argument("--only", default=None, help="substring of the projection names to run")
    ap.add_argument("--skip-vanilla", action="store_true")
    ap.add_argument("--json-output", type=Path)
    args = ap.parse_args(argv)
    if args.warmup < 0 or args.iterations <= 0 or args.repeats <= 0:
        ap.error("--warmup must be non-negative; --iterations and --repeats must be positive")
    return args


def weight_bytes(K: int, N: int) -> float:
    return K * N * 0.53125


def benchmark_comparison(variants, warmup: int, iterations: int, repeats: int) -> dict:
    """variants: [(name, call(i))]; returns medians_us / samples_us / measurement_orders like the reference's record."""
    import torch

    orders = list(permutations([name for name, _ in variants]))
    calls = dict(variants)
    samples = {name: [] for name in calls}
    used_orders = []
    graphs = {}
    side = torch.cuda.Stream()
    for name, call in variants:
        for i in range(warmup):
            call(i)
        torch.cuda.synchronize()
        # the timed calls are captured once and replayed: a Python launch costs more than these kernels run, and a host-bound
        # loop would time the interpreter
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for i in range(iterations):
                    call(i)
        graphs[name] = g
    torch.cuda.synchronize()
    for rep in range(repeats):
        order = orders[rep % len(orders)]
        used_orders.append(list(order))
        for name in order:
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            graphs[name].replay()
            stop.record()
            stop.synchronize()
            samples[name].append(start.elapsed_time(stop) * 1e3 / iterations)
    return {"medians_us": {k: median(v) for k, v in samples.items()}, "samples_us": samples, "measurement_orders": used_orders}


def main(argv=None) -> dict:
    args = parse_args(argv)
    import torch

    import tiny_llm_ext_hip as ext
    from tiny_llm_hip.synthetic import quantize

    if not torch.cuda.is_available():
        raise SystemExit("bench_week2_operators needs a GPU: the extension has no CPU fallback")
    ext.load_library(str(ROOT))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    results = []
    print(f"{'projection':18s} {'K x N':>15s} | " + " | ".join(f"{v:>24s}" for v in ("vanilla us (GB/s)", "operator us (GB/s)", "engine us (GB/s, frac)")))
    for name, K, N in PROJECTIONS:
        if args.only and args.only not in name:
            continue
        nbytes = weight_bytes(K, N)
        copies = max(2, min(64, int((512 << 20) // nbytes) + 1))
        packs, tiled = [], []
        for _ in range(copies):
            w = (torch.randn((K, N), generator=gen, device="cuda", dtype=torch.float32) * 0.02).to(torch.bfloat16)
            packed, scales, biases = quantize(w)
            packs.append((packed, scales, biases))
            tiled.append(ext.TiledW4(packed, scales, biases))
        del w
        x = torch.randn((1, N), generator=gen, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        variants = []
        if not args.skip_vanilla and K * N <= (64 << 20):  # one thread per output over 389 M weights is minutes, not a bench
            variants.append(("vanilla", lambda i: ext.quantized_matmul(packs[i % copies][1], packs[i % copies][2], 128, 4, x,
                                                                      packs[i % copies][0], True, use_simdgroup=False)))
        variants.append(("operator", lambda i: ext.quantized_matmul(packs[i % copies][1], packs[i % copies][2], 128, 4, x,
                                                                  packs[i % copies][0], True)))
        variants.append(("engine", lambda i: ext.decode_linear(tiled[i % copies], x, kernel=1)))
        timing = benchmark_comparison(variants, args.warmup, args.iterations, args.repeats)
        med = timing["medians_us"]
        rate = {k: nbytes / v / 1e3 for k, v in med.items()}
        _, info = ext.decode_linear(tiled[0], x, kernel=1)
        cells = []
        for v in ("vanilla", "operator", "engine"):
            if v not in med:
                cells.append(f"{'-':>24s}")
            elif v == "engine":
                cells.append(f"{med[v]:8.2f} ({rate[v]:6.0f}, {rate[v] / HBM_PEAK_GBPS:.3f})")
            else:
                cells.append(f"{med[v]:11.2f} ({rate[v]:9.0f})")
        print(f"{name:18s} {f'{K} x {N}':>15s} | " + " | ".join(f"{c:>24s}" for c in cells), flush=True)
        results.append({"name": name, "out_features": K, "in_features": N, "weight_bytes": nbytes, "weight_copies": copies,
                        "engine_kernel": info, "GBps": rate, "frac_of_8TBps": {k: v / HBM_PEAK_GBPS for k, v in rate.items()},
                        **timing})
        for t in tiled:
            t.close()
        del packs, tiled
        torch.cuda.empty_cache()
    total = sum(r["weight_bytes"] * (1 if r["name"] == "lm head" else 36) for r in results)
    t_eng = sum(r["medians_us"]["engine"] * (1 if r["name"] == "lm head" else 36) for r in results)
    summary = {"decode_step_weight_bytes": total, "engine_sum_us_per_step": t_eng,
               "engine_GBps_over_step_weights": total / t_eng / 1e3 if t_eng else None}
    if not args.only:
        print(f"one decode step = 36 x (q, k, v, o, gate, up, down) + lm head: {total / 1e9:.3f} GB in {t_eng:.0f} us of separate "
              f"launches -> {summary['engine_GBps_over_step_weights']:.0f} GB/s (the engine fuses q|k|v and gate|up into one launch each)")
    out = {"section": "decode-projections", "warmup": args.warmup, "iterations": args.iterations, "repeats": args.repeats,
           "results": results, "summary": summary}
    if args.json_output:
        args.json_output.parent.mkdir(parents=True, exist_ok=True)
        args.json_output.write_text(json.dumps(out, indent=1))
        print(f"Wrote {args.json_output}")
    return out


if __name__ == "__main__":
    main()
