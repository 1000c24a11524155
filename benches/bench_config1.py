#!/usr/bin/env python3
"""BASELINE config 1: Qwen3-0.6B Week-1 greedy decode on the HOST (plumbing check, no GPU, no extension kernel).

The reference runs this configuration on `mx.cpu`: `Qwen3ModelWeek1` (dense bf16 weights obtained by dequantising the 4-bit
checkpoint, `qwen3_week1.py:206-217`), NO KV cache -- every new token re-runs the whole context -- and the greedy loop
`simple_generate` (`generate.py:16-40`).  Here the same classes of the host mirror (`tiny_llm_hip.Qwen3ModelWeek1`: plain
torch ops) run on CPU tensors.  Synthetic 0.6B-shaped weights (hidden 1024, 28 layers, 16/8 heads of 128, intermediate 3072,
vocab 151,936, tied embeddings); prompt 32 tokens, 16 generated, seed 0 (a short fixed case, the loop is
O(S^2)).  Prints the reference harness's lines; `--layers` shrinks the model for the CPU test tier.

    python benches/bench_config1.py [--layers 28] [--prompt-len 32] [--new-tokens 16] [--threads N]
"""

from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path
from random import Random

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--layers", type=int, default=None, help="override num_hidden_layers (tests)")
    ap.add_argument("--vocab", type=int, default=None, help="override vocab_size (tests)")
    ap.add_argument("--prompt-len", type=int, default=32)
    ap.add_argument("--new-tokens", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--json-output", type=Path)
    args = ap.parse_args(argv)

    import torch

    from tiny_llm_hip import Qwen3ModelWeek1
    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    if args.threads:
        torch.set_num_threads(args.threads)
    cfg = dict(QWEN3_CONFIGS[args.model])
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    if args.vocab:
        cfg["vocab_size"] = args.vocab
    t0 = time.perf_counter()
    model = Qwen3ModelWeek1(synthetic_qwen3(cfg, seed=args.seed, sigma=0.02, device="cpu"))
    build_s = time.perf_counter() - t0
    rng = Random(args.seed)
    low = 256 if cfg["vocab_size"] > 512 else 0
    tokens = [rng.randint(low, cfg["vocab_size"] - 1) for _ in range(args.prompt_len)]
    generated = []
    t0 = time.perf_counter()
    with torch.no_grad():
        logits = model(torch.tensor([tokens], dtype=torch.int32))[:, -1, :].float()  # prefill = the first full pass
        prefill_s = time.perf_counter() - t0
        generated.append(int(torch.argmax(logits, dim=-1)))
        t1 = time.perf_counter()
        for _ in range(args.new_tokens - 1):  # Week 1: no KV cache, the whole context again for every token
            logits = model(torch.tensor([tokens + generated], dtype=torch.int32))[:, -1, :].float()
            generated.append(int(torch.argmax(logits, dim=-1)))
        decode_s = time.perf_counter() - t1
    total = prefill_s + decode_s
    out = {"config": "Qwen3-0.6B Week-1 greedy decode on the host CPU (BASELINE.json configs[0])", "layers": cfg["num_hidden_layers"],
           "threads": torch.get_num_threads(), "prompt_tokens": args.prompt_len, "generated_tokens": len(generated),
           "build_s": build_s, "prefill_tok_s": args.prompt_len / prefill_s, "decode_tok_s": (len(generated) - 1) / decode_s if decode_s else 0.0,
           "output_tok_s": len(generated) / total, "first_ids": generated[:8]}
    print(f"Requests: 1, Prompt tokens: {args.prompt_len}, Generated tokens: {len(generated)}")
    print(f"Time: {total:.2f}s, Output throughput: {out['output_tok_s']:.2f} tok/s")
    print(f"Prefill throughput: {out['prefill_tok_s']:.2f} tok/s")
    print(f"Decode throughput: {out['decode_tok_s']:.2f} tok/s")
    print(f"(host CPU, {out['threads']} torch threads, {cfg['num_hidden_layers']} layers, dense bf16 weights, no KV cache)")
    if args.json_output:
        args.json_output.parent.mkdir(parents=True, exist_ok=True)
        args.json_output.write_text(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
