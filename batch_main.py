#!/usr/bin/env python3
"""Continuous-batching CLI over the fused engine (counterpart of the reference batch-main.py:62-101).

  python batch_main.py --model <checkpoint dir> [--batch-size 5] [--prefill-step 128] [--max-seq-len 512] [--prompts-file f]
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

DEFAULT_PROMPTS = ["What is the capital of France?", "Where is New York City?", "Where is Tokyo?",
                   "What is the capital of China?", "Give me a short introduction to large language model."]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", required=True)
    ap.add_argument("--batch-size", type=int, default=5)
    ap.add_argument("--prefill-step", type=int, default=128)
    ap.add_argument("--max-seq-len", type=int, default=512)
    ap.add_argument("--kv-format", default="bf16", choices=["bf16", "fp8"],
                    help="K / V pages as bfloat16 (the reference's cache) or FP8 E4M3 codes + power-of-two row scales (extension)")
    ap.add_argument("--enable-thinking", action="store_true")
    ap.add_argument("--raw-prompts", action="store_true", help="do not wrap the prompts in the chat template")
    ap.add_argument("--prompts-file", default=None, help="one prompt per line (default: five built-in questions)")
    args = ap.parse_args(argv)

    from tiny_llm_hip import load
    from tiny_llm_hip.engine import DecodeEngine, batch_generate_ids
    from main import chat_prompt

    model, tokenizer = load(args.model)
    prompts = [l for l in Path(args.prompts_file).read_text().splitlines() if l.strip()] if args.prompts_file else DEFAULT_PROMPTS
    encoded, limits = [], []
    for text in prompts:
        full = text if args.raw_prompts else chat_prompt(tokenizer, text, args.enable_thinking)
        ids = tokenizer.encode(full, add_special_tokens=False)
        if len(ids) >= args.max_seq_len:
            raise ValueError(f"prompt of {len(ids)} tokens exceeds max_seq_len {args.max_seq_len}")
        encoded.append(ids)
        limits.append(args.max_seq_len - len(ids))
    pages_per_seq = args.max_seq_len // 128 + 2
    engine = DecodeEngine(model, page_size=128, num_pages=pages_per_seq * (args.batch_size + 1) + 2,
                          max_batch=args.batch_size + 1, max_prefill_rows=args.prefill_step, kv_format=args.kv_format)
    try:
        done = batch_generate_ids(engine, encoded, limits, batch_size=args.batch_size, prefill_step=args.prefill_step,
                                  eos_token_id=tokenizer.eos_token_id)
    finally:
        engine.close()
    results = []
    for idx, ids in done:
        if ids and ids[-1] == tokenizer.eos_token_id:
            ids = ids[:-1]
        text = tokenizer.decode(ids)
        results.append((idx, text))
        print(f"--- {idx} ---\nQ: {prompts[idx]}\nA: {text}")
    return results


if __name__ == "__main__":
    main()
