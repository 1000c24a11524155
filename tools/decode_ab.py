#!/usr/bin/env python3
"""A/B of engine knobs on bench.py's workload inside ONE process (the model is built once; every variant gets its own
engine, whose knobs are read from the environment at tl_engine_create).

    python tools/decode_ab.py [--prompt-len 128] [--steps 256] [--batch 1] VAR=VAL,VAR=VAL ...   (one argument per variant; "-" = defaults)

Prints one JSON line per variant: ms/step over the timed steps (graph replay), and the per-kind kernel time of profiled
steps (in-kernel stamps).  Measurement tool, not part of the product."""
import argparse
import json
import os
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    sys.path.insert(0, str(p))

# every knob the library still reads + the host mirror's TL_ENGINE_OPTIONS ("qmm7=0+aql_fences=1": tl_engine_set_option)
KNOBS = ("TL_AQL", "TL_ATTN_RQ", "TL_ATTN_MAX_SPLITS", "TL_ATTN_MIN_TOKENS", "TL_QMM3_MIN_M", "TL_ATTN_MFMA", "TL_ENGINE_OPTIONS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prefill-step", type=int, default=2048)
    ap.add_argument("--profile-steps", type=int, default=4)
    ap.add_argument("variants", nargs="*", default=["-"])
    args = ap.parse_args()

    import torch
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    cfg = dict(QWEN3_CONFIGS["qwen3-4b"])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda")
    rng = random.Random(0)
    prompts = [[rng.randrange(256, cfg["vocab_size"]) for _ in range(args.prompt_len)] for _ in range(args.batch)]
    page = 128
    per_seq = (args.prompt_len + args.warmup + args.steps + args.profile_steps + 64 + page - 1) // page + 1
    for variant in args.variants:
        for k in KNOBS:
            os.environ.pop(k, None)
        if variant != "-":
            for kv in variant.split(","):
                k, v = kv.split("=", 1)  # (TL_ENGINE_OPTIONS=qmm7=0+aql_fences=1: options joined by "+")
                os.environ[k] = v
        eng = DecodeEngine(model, page_size=page, num_pages=per_seq * args.batch + 2, max_batch=args.batch,
                           max_pages_per_seq=per_seq, max_prefill_rows=max(min(args.prefill_step, args.prompt_len), 8))
        try:
            for i, p in enumerate(prompts):
                eng.begin(i)
                eng.prefill(i, p, chunk=args.prefill_step)
            eng.decode(max(args.warmup, 2), batch=args.batch)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.decode(args.steps, batch=args.batch)
            eng.synchronize()
            dt = time.perf_counter() - t0
            kinds = None
            for _ in range(args.profile_steps):
                p = eng.profile_step(args.batch)
                if kinds is None:
                    kinds = {k: dict(v) for k, v in p["kinds"].items()}
                    n_splits = p["n_splits"]
                else:
                    for k, v in p["kinds"].items():
                        kinds[k]["us"] += v["us"]
            ids = eng.read_tokens(0, 4)
            out = {"variant": variant, "batch": args.batch, "prompt": args.prompt_len, "steps": args.steps,
                   "ms_per_step": round(dt * 1e3 / args.steps, 4), "tokens_per_s": round(args.batch * args.steps / dt, 1),
                   "first_ids": ids}
            if kinds:
                n = args.profile_steps
                out["n_splits"] = n_splits
                out["us_per_step"] = {k: round(v["us"] / n, 1) for k, v in kinds.items()}
                out["launches"] = sum(v["launches"] for v in kinds.values())
                out["kernel_us_per_step"] = round(sum(v["us"] for v in kinds.values()) / n, 1)
            print(json.dumps(out), flush=True)
        finally:
            eng.close()
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
