#!/usr/bin/env python3
"""Lab (DESIGN "Next 1"): does a batched decode step split into two micro-batches overlap on this chip?

Two DecodeEngines on ONE device (each with its own HSA queue on the AQL route, its own weights copy and pages), B / 2
sequences each, decode N steps (a) one after the other and (b) from two host threads at once; against one engine
with B sequences.  The kernels of a 32-row step are latency- / L2->CU-bound, not HBM-bound, so two independent
chains could share the chip.  Prints one JSON line per B.  Not a product path.

  python tools/lab/two_engines_probe.py --batches 16,32,64 --steps 32
"""
import argparse
import json
import pathlib
import random
import sys
import threading
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tiny-llm_amd"))
sys.path.insert(0, str(ROOT / "tiny-llm_amd" / "extensions_hip"))


def make_engine(model, cfg, batch, context, steps, seed):
    from tiny_llm_hip.engine import DecodeEngine
    page = 128
    per_seq = (context + 3 * steps + 2 * page) // page + 1
    e = DecodeEngine(model, page_size=page, num_pages=per_seq * batch + 2, max_batch=batch, max_prefill_rows=128)
    rng = random.Random(seed)
    for slot in range(batch):
        e.begin(slot)
        e.prefill(slot, [rng.randrange(256, cfg["vocab_size"]) for _ in range(context)], chunk=128)
    e.decode(4, batch=batch)
    e.synchronize()
    return e


def timed(fn):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,32,64")
    ap.add_argument("--context", type=int, default=128)
    ap.add_argument("--steps", type=int, default=32)
    args = ap.parse_args()
    import torch
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3
    cfg = dict(QWEN3_CONFIGS["qwen3-4b"])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
    for B in [int(x) for x in args.batches.split(",")]:
        h = B // 2
        whole = make_engine(model, cfg, B, args.context, args.steps, 0)
        a = make_engine(model, cfg, h, args.context, args.steps, 1)
        b = make_engine(model, cfg, B - h, args.context, args.steps, 2)
        n = args.steps

        def run(e, bb):
            e.decode(n, batch=bb)
            e.synchronize()

        t_whole = timed(lambda: run(whole, B))
        t_seq = timed(lambda: (run(a, h), run(b, B - h)))

        def both():
            ts = [threading.Thread(target=run, args=(a, h)), threading.Thread(target=run, args=(b, B - h))]
            for t in ts:
                t.start()
            for t in ts:
                t.join()

        t_par = timed(both)
        t_half = timed(lambda: run(a, h))
        print(json.dumps({"sequences": B, "route": whole.replay_route(),
                          "one_engine_ms_per_step": round(t_whole * 1e3 / n, 4),
                          "half_alone_ms_per_step": round(t_half * 1e3 / n, 4),
                          "two_halves_back_to_back_ms_per_step_pair": round(t_seq * 1e3 / n, 4),
                          "two_halves_concurrent_ms_per_step_pair": round(t_par * 1e3 / n, 4)}), flush=True)
        for e in (whole, a, b):
            e.close()
        del whole, a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
