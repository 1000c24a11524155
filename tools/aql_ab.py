#!/usr/bin/env python3
"""A/B of the decode step's two replay routes on one box: hipGraphLaunch (TL_AQL=0) against AQL packets on the engine's own HSA queue
(the default, csrc/aql.h; aql_fences = the same packets with HIP's agent-scope fences on every one) -- same captured step, same kernels.  Prints ms per step, the greedy ids (must be identical) and the engine's counters.

  python tools/aql_ab.py [--prompt 128] [--steps 64] [--rounds 3] [--modes graph,aql,aql_fences]
"""
import argparse
import json
import os
import pathlib
import random
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tiny-llm_amd"))
sys.path.insert(0, str(ROOT / "tiny-llm_amd" / "extensions_hip"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--modes", default="graph,aql")
    ap.add_argument("--model", default="qwen3-4b")
    args = ap.parse_args()

    import torch
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    cfg = dict(QWEN3_CONFIGS[args.model])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
    rng = random.Random(0)
    prompt = [rng.randrange(256, cfg["vocab_size"]) for _ in range(args.prompt)]
    env = {"graph": {"TL_AQL": "0"}, "aql": {"TL_AQL": "1"}, "aql_fences": {"TL_AQL": "1", "TL_ENGINE_OPTIONS": "aql_fences=1"},
           "default": {}}
    results = {}
    for rnd in range(args.rounds):
        for mode in args.modes.split(","):
            for k in ("TL_AQL", "TL_ENGINE_OPTIONS"):
                os.environ.pop(k, None)
            os.environ.update(env[mode])
            eng = DecodeEngine(model, page_size=128, num_pages=(args.prompt + args.steps + 80) // 128 + 3, max_batch=1, max_prefill_rows=128)
            try:
                eng.begin(0)
                eng.prefill(0, prompt, chunk=128)
                eng.decode(8, batch=1)
                eng.synchronize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.decode(args.steps, batch=1)
                eng.synchronize()
                dt = time.perf_counter() - t0
                ids = eng.read_tokens(0, args.steps + 9)
                logits = eng.logits(1)[0].float().cpu()
                st = eng.stats()
                eng.release(0)
            finally:
                eng.close()
            r = results.setdefault(mode, {"ms_per_step": [], "ids": ids, "logits": logits})
            r["ms_per_step"].append(round(dt * 1e3 / args.steps, 4))
            r["aql_steps"] = st.get("aql_steps")
            r["graph_replays"] = st["graph_replays"]
    base = results[args.modes.split(",")[0]]
    for mode, r in results.items():
        same_ids = r["ids"] == base["ids"]
        diff = float((r["logits"] - base["logits"]).abs().max())
        print(json.dumps({"mode": mode, "ms_per_step": r["ms_per_step"], "best": min(r["ms_per_step"]), "aql_steps": r["aql_steps"],
                          "graph_replays": r["graph_replays"], "ids_equal_first_mode": same_ids, "max_abs_logit_diff_vs_first_mode": diff,
                          "first_ids": r["ids"][:6]}))


if __name__ == "__main__":
    main()
