#!/usr/bin/env python3
"""Model-level throughput harness with the reference's CLI shape and output lines
(reference: benches/bench.py — flags :64-132, request generator :190-225, single-request loop :277-312, continuous
batching loop :351-572, report :769-800; drivers parse the "(Prefill|Decode|Output) throughput: X tok/s" lines,
benches/bench_course_progression.py:103-105).

    python -m benches.bench --model qwen3-4b --num-seqs 1 --min-input-len 128 --max-input-len 128 \
        --min-output-len 129 --max-output-len 129 --prefill-logits last          # reference acceptance shape
    python -m benches.bench --batch-decode --batch-size 64 --num-seqs 128 --min-input-len 128 --max-input-len 1024 \
        --min-output-len 32 --max-output-len 128 --prefill-step 128 --prefill-budget 2048   # serving trace (config 4, one GPU;
                                                                  # benches/serve_replicas.py deals it over N GPUs)

--solution engine (default): the fused decode engine (tinyllm_engine.h).  --solution ops: the op-by-op
``Qwen3ModelWeek3`` on the HIP operators, i.e. the reference's own call structure.  Weights are synthetic
(random-init, the checkpoint cannot be downloaded here); token ids come from the reference's generator.
"""

from __future__ import annotations

import argparse
import json
import sys
import time
from dataclasses import asdict, dataclass
from pathlib import Path
from random import Random

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


@dataclass
class BenchRequest:
    prompt_token_ids: list[int]
    max_new_tokens: int


def random_token_id(rng: Random, low: int, high: int, eos_token_id: int) -> int:
    if low == high:
        return low
    token = rng.randint(low, high)
    if token != eos_token_id:
        return token
    return low + 1 if token == low else token - 1


def build_requests(*, rng: Random, num_seqs: int, vocab_size: int, eos_token_id: int, min_input_len: int,
                   max_input_len: int, min_output_len: int, max_output_len: int) -> list[BenchRequest]:
    """Same draw order as the reference generator, so a seed reproduces the reference's trace."""
    token_low = 256 if vocab_size > 512 else 0
    token_high = vocab_size - 1
    if token_low > token_high:
        token_low = 0
    requests = []
    for _ in range(num_seqs):
        prompt_len = rng.randint(min_input_len, max_input_len)
        max_new_tokens = rng.randint(min_output_len, max_output_len)
        prompt = [random_token_id(rng, token_low, token_high, eos_token_id) for _ in range(prompt_len)]
        requests.append(BenchRequest(prompt, max_new_tokens))
    return requests


def safe_div(a: float, b: float) -> float:
    return a / b if b else 0.0


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--solution", default="engine", choices=["engine", "ops"])
    ap.add_argument("--num-seqs", type=int, default=16)
    ap.add_argument("--min-input-len", type=int, default=64)
    ap.add_argument("--max-input-len", type=int, default=256)
    ap.add_argument("--min-output-len", type=int, default=64)
    ap.add_argument("--max-output-len", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--prefill-logits", choices=("all", "last"), default="last")
    ap.add_argument("--batch-decode", action="store_true")
    ap.add_argument("--batch-size", type=int, default=5)
    ap.add_argument("--prefill-step", type=int, default=128)
    ap.add_argument("--prefill-budget", type=int, default=None,
                    help="prompt tokens a serving turn may spend on admission before its decode step (default: one "
                         "--prefill-step chunk, the reference's schedule; config 4 uses 2048 so that 64 slots fill)")
    ap.add_argument("--staging-slots", type=int, default=1,
                    help="--batch-decode: prompts prefilled together per turn (1 = one at a time as the reference; > 1 packs them "
                         "into one multi-token pass of at most --prefill-budget rows)")
    ap.add_argument("--keep-slot-holes", action="store_true",
                    help="do not move live requests into the decode slots finished requests left (benches/serving.py _close_holes): the step then "
                         "decodes the prefix up to the highest live slot, as before round 5")
    ap.add_argument("--page-size", type=int, default=128)
    ap.add_argument("--kv-format", default="bf16", choices=["bf16", "fp8"],
                    help="--solution engine: K / V pages as bfloat16 (the reference's cache) or FP8 E4M3 codes + power-of-two row scales (extension; no reference behaviour)")
    ap.add_argument("--json-output", type=Path)
    args = ap.parse_args(argv)
    if args.num_seqs <= 0 or args.batch_size <= 0 or args.prefill_step <= 0:
        raise ValueError("--num-seqs, --batch-size and --prefill-step must be > 0")
    if args.min_input_len <= 0 or args.max_input_len < args.min_input_len:
        raise ValueError("input length range is empty")
    if args.min_output_len <= 0 or args.max_output_len < args.min_output_len:
        raise ValueError("output length range is empty")
    return args


# ------------------------------------------------------------------------------------------------ engine runners
def run_one_request_engine(engine, request: BenchRequest, prefill_step: int):
    """Reference run_one_request_week2: one prefill (timed with its sync), then one sync per decoded token."""
    engine.begin(0)
    try:
        t0 = time.perf_counter()
        engine.prefill(0, request.prompt_token_ids, chunk=prefill_step)
        engine.synchronize()
        prefill_time = time.perf_counter() - t0
        t1 = time.perf_counter()
        for _ in range(request.max_new_tokens - 1):
            engine.decode(1, batch=1)
            engine.synchronize()  # the reference evaluates every token (mx.eval) inside the timer
        decode_time = time.perf_counter() - t1
        return request.max_new_tokens, prefill_time, decode_time
    finally:
        engine.release(0)


# ------------------------------------------------------------------------------------------------ op-by-op runner
def run_one_request_ops(model, request: BenchRequest, prefill_step: int):
    import torch

    cache = model.create_kv_cache()
    try:
        prompt = torch.tensor([request.prompt_token_ids], dtype=torch.int32, device="cuda")
        t0 = time.perf_counter()
        offset, token = 0, None
        while offset < prompt.shape[1]:
            part = prompt[:, offset:offset + prefill_step]
            logits = model(part, offset, cache, logits_to_keep=1)
            offset += part.shape[1]
            token = torch.argmax(logits[:, -1], dim=-1)
        torch.cuda.synchronize()
        prefill_time = time.perf_counter() - t0
        t1 = time.perf_counter()
        for _ in range(request.max_new_tokens - 1):
            logits = model(token.to(torch.int32)[None], offset, cache, logits_to_keep=1)
            token = torch.argmax(logits[:, -1], dim=-1)
            torch.cuda.synchronize()
            offset += 1
        decode_time = time.perf_counter() - t1
        return request.max_new_tokens, prefill_time, decode_time
    finally:
        for layer_cache in cache:
            layer_cache.release()


def main(argv=None) -> None:
    args = parse_args(argv)
    import torch

    from tiny_llm_hip.synthetic import QWEN3_CONFIGS, synthetic_qwen3

    if not torch.cuda.is_available():
        raise SystemExit("benches.bench needs a GPU: the HIP path has no CPU fallback")
    cfg = dict(QWEN3_CONFIGS[args.model])
    mlx_model = synthetic_qwen3(cfg, seed=args.seed, sigma=0.02, device="cuda")
    eos = cfg["vocab_size"] - 1  # never sampled by the generator below, never matched by the greedy loop on purpose
    requests = build_requests(rng=Random(args.seed), num_seqs=args.num_seqs, vocab_size=cfg["vocab_size"], eos_token_id=eos,
                              min_input_len=args.min_input_len, max_input_len=args.max_input_len,
                              min_output_len=args.min_output_len, max_output_len=args.max_output_len)
    total_prompt = sum(len(r.prompt_token_ids) for r in requests)
    longest = max(len(r.prompt_token_ids) + r.max_new_tokens for r in requests)
    metrics = None

    if args.solution == "engine":
        from tiny_llm_hip.engine import DecodeEngine

        slots = args.batch_size + max(1, args.staging_slots) if args.batch_decode else 1
        pages_per_seq = (longest + args.page_size - 1) // args.page_size + 1
        engine = DecodeEngine(mlx_model, page_size=args.page_size, num_pages=pages_per_seq * slots + 2, max_batch=slots,
                              max_pages_per_seq=pages_per_seq,
                              max_prefill_rows=max(args.prefill_step, (args.prefill_budget or 0) if args.staging_slots > 1 else 0, 8),
                              kv_format=args.kv_format)

        kv_row_bytes = cfg["head_dim"] * 2 if args.kv_format == "bf16" else cfg["head_dim"] + 4
        kv_page_bytes = 2 * cfg["num_hidden_layers"] * cfg["num_key_value_heads"] * args.page_size * kv_row_bytes

        def run_all(reqs):
            nonlocal metrics
            if args.batch_decode:
                from benches.serving import serve_requests

                metrics = serve_requests(engine, reqs, batch_size=args.batch_size, prefill_step=args.prefill_step,
                                         prefill_budget=args.prefill_budget, page_size=args.page_size,
                                         kv_bytes_per_page=kv_page_bytes, capacity_pages=pages_per_seq * slots + 2,
                                         staging_slots=args.staging_slots, compact=not args.keep_slot_holes)
                return metrics.generated_tokens, metrics.decode_tokens, metrics.prefill_time, metrics.decode_time
            gen = dec = 0
            pt = dt = 0.0
            for r in reqs:
                g, p, d = run_one_request_engine(engine, r, args.prefill_step)
                gen, dec, pt, dt = gen + g, dec + max(0, g - 1), pt + p, dt + d
            return gen, dec, pt, dt
    else:
        from tiny_llm_hip import Qwen3ModelWeek3

        if args.batch_decode:
            raise SystemExit("--solution ops supports the static (one request at a time) mode only")
        model = Qwen3ModelWeek3(mlx_model, page_size=args.page_size)

        def run_all(reqs):
            gen = dec = 0
            pt = dt = 0.0
            for r in reqs:
                g, p, d = run_one_request_ops(model, r, args.prefill_step)
                gen, dec, pt, dt = gen + g, dec + max(0, g - 1), pt + p, dt + d
            return gen, dec, pt, dt

    for _ in range(args.warmup):  # complete-request warmups, like the reference
        run_all(requests[: max(1, min(len(requests), args.batch_size if args.batch_decode else 1))])
    t0 = time.perf_counter()
    generated, decode_tokens, prefill_time, decode_time = run_all(requests)
    total_time = time.perf_counter() - t0

    payload = {"config": vars(args) | {"json_output": str(args.json_output) if args.json_output else None},
               "metrics": {"requests": args.num_seqs, "prompt_tokens": total_prompt, "generated_tokens": generated,
                           "total_time_s": total_time, "output_tok_s": safe_div(generated, total_time),
                           "prefill_tok_s": safe_div(total_prompt, prefill_time),
                           "decode_tok_s": safe_div(decode_tokens, decode_time)}}
    if metrics is not None:
        from benches.serving import report_lines

        print("\n".join(report_lines(args.num_seqs, total_prompt, total_time, metrics)))
        serving = asdict(metrics)
        serving.pop("decode_step_ms")
        payload["metrics"] |= {"req_s": safe_div(args.num_seqs, total_time), **serving}
    else:
        print(f"Requests: {args.num_seqs}, Prompt tokens: {total_prompt}, Generated tokens: {generated}")
        print(f"Time: {total_time:.2f}s, Output throughput: {safe_div(generated, total_time):.2f} tok/s")
        print(f"Total throughput (prompt+output): {safe_div(total_prompt + generated, total_time):.2f} tok/s")
        print(f"Prefill throughput: {safe_div(total_prompt, prefill_time):.2f} tok/s")
        print(f"Decode throughput: {safe_div(decode_tokens, decode_time):.2f} tok/s")
    if args.json_output:
        args.json_output.parent.mkdir(parents=True, exist_ok=True)
        args.json_output.write_text(json.dumps(payload, indent=1, default=str))


if __name__ == "__main__":
    main()
