#!/usr/bin/env python3
"""Per-kind in-kernel time of a decode step at several sequence counts, ONE model build (synthetic Qwen3-4B W4): one JSON line per count.

  python tools/batch_profile_sweep.py 8 16 24 32 48 64 [--context 128] [--steps 24]
"""
import argparse
import json
import pathlib
import random
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tiny-llm_amd"))
sys.path.insert(0, str(ROOT / "tiny-llm_amd" / "extensions_hip"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("batches", type=int, nargs="+")
    ap.add_argument("--context", type=int, default=128)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--env", action="append", default=[], help="KEY=VAL set around every engine (read at tl_engine_create); repeat for several; the line carries them")
    args = ap.parse_args()
    import os
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import torch
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    cfg = dict(QWEN3_CONFIGS["qwen3-4b"])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
    rng = random.Random(0)
    page = 128
    for b in args.batches:
        per_seq = (args.context + args.steps + 8 + 2 * page) // page + 1
        eng = DecodeEngine(model, page_size=page, num_pages=per_seq * b + 2, max_batch=b, max_prefill_rows=128)
        for slot in range(b):
            eng.begin(slot)
            eng.prefill(slot, [rng.randrange(256, cfg["vocab_size"]) for _ in range(args.context)], chunk=128)
        eng.decode(4, batch=b)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.decode(args.steps, batch=b)
        eng.synchronize()
        dt = time.perf_counter() - t0
        p = eng.profile_step(b)
        kinds = {k: [round(v["us"], 1), v["launches"]] for k, v in p["kinds"].items() if v["launches"]}
        print(json.dumps({"batch": b, "env": args.env, "context": args.context, "ms_per_step": round(dt / args.steps * 1e3, 4), "route": eng.replay_route(),
                          "n_splits": p.get("n_splits"), "kernel_us": round(sum(v[0] for v in kinds.values()), 1), "kinds_us_launches": kinds}), flush=True)
        for slot in range(b):
            eng.release(slot)
        eng.close()


if __name__ == "__main__":
    main()
