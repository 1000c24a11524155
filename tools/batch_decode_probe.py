#!/usr/bin/env python3
"""Decode-step latency of the fused engine at a given number of concurrent sequences (synthetic Qwen3-4B W4).

  python tools/batch_decode_probe.py --batch 64 --context 256 --steps 32
  rocprofv3 --kernel-trace --stats ... -- python tools/batch_decode_probe.py --batch 64 --no-graph --steps 8

Prints one JSON line: ms per step, aggregate tokens/s, algorithmic bytes per step and GB/s (SURVEY.md §8d:
W + 147,456 B x sum of contexts).  Not a product path; used to size the batched-decode kernels.
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--context", type=int, default=256, help="tokens already in every sequence when timing starts")
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile", action="store_true", help="also print the per-kind in-kernel stamps of one profiled step (tl_engine_profile_step)")
    args = ap.parse_args()

    import torch
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    cfg = dict(QWEN3_CONFIGS[args.model])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
    page = 128
    per_seq = (args.context + args.warmup + args.steps + 2 * page) // page + 1
    engine = DecodeEngine(model, page_size=page, num_pages=per_seq * args.batch + 2, max_batch=args.batch,
                          max_prefill_rows=128)
    rng = random.Random(0)
    for slot in range(args.batch):
        engine.begin(slot)
        engine.prefill(slot, [rng.randrange(256, cfg["vocab_size"]) for _ in range(args.context)], chunk=128)
    engine.decode(max(args.warmup, 2), batch=args.batch, use_graph=not args.no_graph)
    engine.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.decode(args.steps, batch=args.batch, use_graph=not args.no_graph)
    engine.synchronize()
    dt = time.perf_counter() - t0
    step_bytes = engine.step_bytes(args.batch)
    prof = None
    if args.profile:
        p = engine.profile_step(args.batch)
        prof = {k: {"us": round(v["us"], 1), "launches": v["launches"]} for k, v in p["kinds"].items() if v["launches"]}
        prof["span_us"] = round(p["span_us"], 1)
        prof["n_splits"] = p.get("n_splits")
    print(json.dumps({"profile": prof,"batch": args.batch, "context": args.context, "steps": args.steps,
                      "ms_per_step": round(dt * 1e3 / args.steps, 4),
                      "tokens_per_s": round(args.batch * args.steps / dt, 1),
                      "step_bytes": int(step_bytes), "step_GBps": round(step_bytes / (dt / args.steps) / 1e9, 1)}))


if __name__ == "__main__":
    main()
