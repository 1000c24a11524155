#!/usr/bin/env python3
"""Lab: the reference's serving trace at 64 slots (bench.py serving64_leg's trace and engine) under several admission budgets / chunk sizes / staging slots.
  python tools/lab/serving_budget_ab.py "512,2048,8" "1024,4096,8" "1024,2048,8" "512,4096,16"      (prefill_step, prefill_budget, staging_slots)
Prints one JSON line per setting.  Not a product path."""
import json
import pathlib
import sys
import time
from random import Random

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tiny-llm_amd"))
sys.path.insert(0, str(ROOT / "tiny-llm_amd" / "extensions_hip"))


def main() -> None:
    import torch
    from benches.bench import build_requests
    from benches.serving import serve_requests, median
    from tiny_llm_hip.engine import DecodeEngine
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    settings = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(512, 2048, 8)]
    cfg = dict(QWEN3_CONFIGS["qwen3-4b"])
    model = synthetic_qwen3(cfg, seed=0, sigma=0.02, device="cuda:0")
    page, B = 128, 64
    trace = build_requests(rng=Random(0), num_seqs=128, vocab_size=cfg["vocab_size"], eos_token_id=cfg["vocab_size"] - 1,
                           min_input_len=128, max_input_len=1024, min_output_len=32, max_output_len=128)
    prompt_tokens = sum(len(r.prompt_token_ids) for r in trace)
    longest = max(len(r.prompt_token_ids) + r.max_new_tokens for r in trace)
    pages_per_seq = (longest + page - 1) // page + 1
    kv_page_bytes = 2 * cfg["num_hidden_layers"] * cfg["num_key_value_heads"] * page * cfg["head_dim"] * 2
    for step, budget, staging in settings:
        slots = B + staging
        eng = DecodeEngine(model, page_size=page, num_pages=pages_per_seq * slots + 2, max_batch=slots, max_pages_per_seq=pages_per_seq, max_prefill_rows=max(2048, budget))

        def run(reqs):
            return serve_requests(eng, reqs, batch_size=B, page_size=page, kv_bytes_per_page=kv_page_bytes, capacity_pages=pages_per_seq * slots + 2,
                                  prefill_step=step, prefill_budget=budget, staging_slots=staging)
        run(trace[:32])
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = run(trace)
        eng.synchronize()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(json.dumps({"prefill_step": step, "prefill_budget": budget, "staging_slots": staging, "wall_s": round(wall, 3),
                          "total_tok_s": round((prompt_tokens + m.generated_tokens) / wall, 1), "output_tok_s": round(m.generated_tokens / wall, 1),
                          "decode_tok_s": round(m.decode_tokens / m.decode_time, 1) if m.decode_time else None,
                          "prefill_tok_s": round(prompt_tokens / m.prefill_time, 1) if m.prefill_time else None,
                          "step_p50_ms": round(median(m.decode_step_ms), 3) if m.decode_step_ms else None, "decode_steps": m.decode_step_count}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
