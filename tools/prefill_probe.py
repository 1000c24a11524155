#!/usr/bin/env python3
"""Chunked-prefill throughput of the fused engine (synthetic Qwen3-4B W4): tokens/s and model TFLOP/s.

  python tools/prefill_probe.py --prompt 8192 --chunk 2048
Model flops per token: 2 x 3,633,315,840 (projections) + causal attention 2*2*Hq*D*S/2 per layer.
Not a product path; used to size the prefill kernels.
"""
import argparse
import json
import pathlib
import random
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tiny-llm_amd"))
import os  # noqa: E402
sys.path.insert(0, os.environ.get("TL_EXT_ROOT") or str(ROOT / "tiny-llm_amd" / "extensions_hip"))  # (an A/B against another build of the library)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=8192)
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()

    import torch
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    cfg = dict(QWEN3_CONFIGS[args.model])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
    page = 128
    engine = DecodeEngine(model, page_size=page, num_pages=args.prompt // page + 4, max_batch=1,
                          max_prefill_rows=args.chunk)
    rng = random.Random(0)
    prompt = [rng.randrange(256, cfg["vocab_size"]) for _ in range(args.prompt)]
    best = None
    for _ in range(args.repeat + 1):  # first pass warms up (workspace growth, code objects)
        engine.begin(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.prefill(0, prompt, chunk=args.chunk)
        engine.synchronize()
        dt = time.perf_counter() - t0
        engine.release(0)
        best = dt if best is None else min(best, dt)
    proj = 2 * 3633315840 * args.prompt
    attn = cfg["num_hidden_layers"] * 2 * 2 * cfg["num_attention_heads"] * cfg["head_dim"] * args.prompt * args.prompt / 2
    print(json.dumps({"prompt": args.prompt, "chunk": args.chunk, "seconds": round(best, 4),
                      "tokens_per_s": round(args.prompt / best, 1), "model_TFLOPs": round((proj + attn) / best / 1e12, 1)}))


if __name__ == "__main__":
    main()
