#!/usr/bin/env python3
"""BASELINE config 4: continuous batching of one request trace over N data-parallel replicas, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        benches/serve_replicas.py --num-seqs 512 --batch-size 64 --min-input-len 128 --max-input-len 1024 \
        --min-output-len 32 --max-output-len 128 --prefill-step 512 --prefill-budget 2048 --staging-slots 8

Admission: up to ``--staging-slots`` prompts are staged at once and their next chunks (``--prefill-step`` rows each, at most
``--prefill-budget`` rows together) go through ONE packed multi-token pass per turn (tl_engine_prefill_packed) -- the W4 GEMM
is efficient from ~1k rows, a lone 128-row chunk is not (r02, 128 requests of 100-1,024 prompt tokens on one GPU: total
throughput 16.7k tok/s with the reference's one-prompt-at-a-time policy (--staging-slots 1 --prefill-step 128), 33.6k with
the defaults).

Every rank builds the SAME seeded trace (reference generator, benches/bench.py:190-225), serves requests ``i mod N``
with its own weight copy, page pools and scheduler (benches/serving.py = the per-replica loop of the reference,
benches/bench.py:351-572), and rank 0 prints the aggregate and per-GPU lines.  Requests shard by index, so there is NO
collective on the data path; the control plane (one barrier before the timed region, one gather of the
per-replica reports after it) runs over gloo on the host -- RCCL/xGMI are not touched at all.

Job throughput = tokens of ALL replicas / the SLOWEST replica's wall time (the job is done when the last one is).
``--solution schedule-only`` runs the same dealer and scheduler against benches.serving.ScheduleOnlyEngine (a cost
model, no GPU): capacity planning, and the CPU test of this file (tests/test_serve_replicas_cpu.py, world size 2).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from dataclasses import asdict
from pathlib import Path
from random import Random

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

from benches.bench import build_requests  # noqa: E402
from benches.serving import ScheduleOnlyEngine, nearest_rank, median, report_lines, serve_requests  # noqa: E402


def deal(requests: list, rank: int, world: int) -> list:
    """Request i goes to replica i mod N: independent of arrival timing, reproducible."""
    return requests[rank::world]


def aggregate(reports: list[dict]) -> dict:
    """Whole-job numbers from the per-replica reports (each: rank, requests, prompt_tokens, wall_s, metrics dict + step list)."""
    wall = max(r["wall_s"] for r in reports)
    gen = sum(r["metrics"]["generated_tokens"] for r in reports)
    dec = sum(r["metrics"]["decode_tokens"] for r in reports)
    prompt = sum(r["prompt_tokens"] for r in reports)
    steps = [ms for r in reports for ms in r["decode_step_ms"]]
    return {
        "replicas": len(reports),
        "requests": sum(r["requests"] for r in reports),
        "prompt_tokens": prompt,
        "generated_tokens": gen,
        "wall_s": wall,
        "output_tok_s": gen / wall if wall else 0.0,
        "total_tok_s": (prompt + gen) / wall if wall else 0.0,
        "req_s": sum(r["requests"] for r in reports) / wall if wall else 0.0,
        # per-replica decode rates add up (each replica times only its own decode steps)
        "decode_tok_s": sum(r["metrics"]["decode_tokens"] / r["metrics"]["decode_time"] for r in reports
                            if r["metrics"]["decode_time"] > 0),
        "prefill_tok_s": sum(r["prompt_tokens"] / r["metrics"]["prefill_time"] for r in reports
                             if r["metrics"]["prefill_time"] > 0),
        "decode_tokens": dec,
        # algorithmic HBM bytes of all decode steps (W + K/V of the live contexts, every step) over the summed decode time: GB/s per GPU
        "decode_bytes": sum(r["metrics"].get("decode_bytes", 0) for r in reports),
        "decode_bytes_per_step": (sum(r["metrics"].get("decode_bytes", 0) for r in reports) / len(steps)) if steps else 0.0,
        "decode_GBps_per_gpu": [r["metrics"].get("decode_bytes", 0) / r["metrics"]["decode_time"] / 1e9 if r["metrics"]["decode_time"] > 0 else 0.0
                                for r in reports],
        "decode_step_p50_ms": median(steps),
        "decode_step_p95_ms": nearest_rank(steps, 0.95),
        "peak_active_requests": [r["metrics"]["peak_active_requests"] for r in reports],
        "slowest_over_fastest_wall": wall / min(r["wall_s"] for r in reports) if reports else 0.0,
        "per_replica": [{"rank": r["rank"], "requests": r["requests"], "request_indices": r.get("request_indices"), "wall_s": r["wall_s"],
                         "output_tok_s": r["metrics"]["generated_tokens"] / r["wall_s"] if r["wall_s"] else 0.0,
                         "decode_tok_s": (r["metrics"]["decode_tokens"] / r["metrics"]["decode_time"]
                                          if r["metrics"]["decode_time"] else 0.0),
                         "decode_step_p50_ms": median(r["decode_step_ms"]),
                         "decode_step_p95_ms": nearest_rank(r["decode_step_ms"], 0.95),
                         "peak_active_requests": r["metrics"]["peak_active_requests"]} for r in reports],
    }


HBM_PEAK_GBPS = 8000.0


def driver_line(args, out: dict, world: int) -> dict:
    """The run as ONE JSON line in the shape of bench.py's (metric, value = whole-job aggregate, n_gpus, config.workload =
    BASELINE.json configs[3]) so that config 4 has a driver-readable record; per-GPU and aggregate rates, bytes per step."""
    gbps = out["decode_GBps_per_gpu"]
    return {
        "metric": "Qwen3-4B int4 decode tokens/sec/GPU; achieved HBM GB/s vs roofline",
        "value": round(out["output_tok_s"], 2), "unit": "tokens/s", "n_gpus": world, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
        "dtype_note": "int4 (W4A16, group 128) weights x bf16 activations on bf16 MFMA, fp32 accumulate",
        "data": ("none (--solution schedule-only: cost model, NOT a measurement)" if args.solution != "engine" else
                 "synthetic (random-init Qwen3-4B-shaped W4 weights, the reference's seeded request trace)"),
        "config": {"workload": "Qwen3-4B continuous batching, 64 concurrent requests, request-parallel across 8xMI355X (BASELINE.json configs[3])",
                   "num_seqs": args.num_seqs, "decode_slots_per_gpu": args.batch_size, "input_len": [args.min_input_len, args.max_input_len],
                   "output_len": [args.min_output_len, args.max_output_len], "prefill_step": args.prefill_step,
                   "prefill_budget": args.prefill_budget, "staging_slots": args.staging_slots, "seed": args.seed,
                   "parallelism": f"request-parallel x{world} (request i -> GPU i mod N, no collective on the data path)"},
        "wall_s": round(out["wall_s"], 4), "requests": out["requests"], "req_per_s": round(out["req_s"], 3),
        "output_tokens_per_s": round(out["output_tok_s"], 2), "total_tokens_per_s": round(out["total_tok_s"], 2),
        "decode_tokens_per_s": round(out["decode_tok_s"], 2), "prefill_tokens_per_s": round(out["prefill_tok_s"], 2),
        "decode_step_p50_ms": round(out["decode_step_p50_ms"], 4), "decode_step_p95_ms": round(out["decode_step_p95_ms"], 4),
        "per_gpu": [{"rank": r["rank"], "requests": r["requests"], "output_tokens_per_s": round(r["output_tok_s"], 2),
                     "decode_tokens_per_s": round(r["decode_tok_s"], 2), "decode_step_p50_ms": round(r["decode_step_p50_ms"], 4)}
                    for r in out["per_replica"]],
        "slowest_over_fastest_wall": round(out["slowest_over_fastest_wall"], 4),
        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                     "bytes_per_decode_step": int(out["decode_bytes_per_step"]),
                     "bytes_rule": "W + 147,456 B x sum of live contexts per step (SURVEY.md section 8d); with --kv-format fp8: 76,032 B (128 codes + a 4-byte scale per row)",
                     "achieved_per_gpu": [round(g, 1) for g in gbps],
                     "achieved": round(sum(gbps) / len(gbps), 1) if gbps else None,
                     "frac": round(sum(gbps) / len(gbps) / HBM_PEAK_GBPS, 4) if gbps else None,
                     "timing": "host clock around each synchronised decode step (the reference's discipline, benches/bench.py:502-522)"},
    }


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--solution", default="engine", choices=["engine", "schedule-only"])
    ap.add_argument("--num-seqs", type=int, default=512)
    ap.add_argument("--min-input-len", type=int, default=128)
    ap.add_argument("--max-input-len", type=int, default=1024)
    ap.add_argument("--min-output-len", type=int, default=32)
    ap.add_argument("--max-output-len", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--batch-size", type=int, default=64, help="decode slots PER REPLICA")
    ap.add_argument("--prefill-step", type=int, default=512, help="rows of one prompt per prefill pass")
    ap.add_argument("--prefill-budget", type=int, default=2048)
    ap.add_argument("--staging-slots", type=int, default=8,
                    help="prompts prefilled together per turn (1 = the reference's one-at-a-time admission; > 1 packs the admitted "
                         "prompts' chunks into one multi-token pass of at most --prefill-budget rows)")
    ap.add_argument("--keep-slot-holes", action="store_true",
                    help="do not move live requests into the decode slots finished requests left (benches/serving.py _close_holes): the step then "
                         "decodes the prefix up to the highest live slot, as before round 5")
    ap.add_argument("--page-size", type=int, default=128)
    ap.add_argument("--kv-format", default="bf16", choices=["bf16", "fp8"],
                    help="K / V pages as bfloat16 (the reference's cache) or FP8 E4M3 codes + power-of-two row scales (extension; no reference behaviour)")
    ap.add_argument("--warmup-requests", type=int, default=None, help="requests of an untimed warm-up pass (default: batch size)")
    ap.add_argument("--json-output", type=Path)
    ap.add_argument("--gpus", type=int, default=None,
                    help="replicas; started without a launcher (WORLD_SIZE unset) and N > 1, this script starts its own N ranks "
                         "under torch.distributed.run on 127.0.0.1")
    return ap.parse_args(argv)


def self_launch(argv: list[str], n: int) -> None:
    """One rank per GPU under torch.distributed.run; rank 0's report goes to this process's stdout."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main(argv=None) -> dict | None:
    args = parse_args(argv)
    if args.gpus and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(list(sys.argv[1:] if argv is None else argv), args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")  # control plane only; the data path has no collective

    from tiny_llm_hip.synthetic import QWEN3_CONFIGS

    cfg = dict(QWEN3_CONFIGS[args.model])
    eos = cfg["vocab_size"] - 1
    trace = build_requests(rng=Random(args.seed), num_seqs=args.num_seqs, vocab_size=cfg["vocab_size"], eos_token_id=eos,
                           min_input_len=args.min_input_len, max_input_len=args.max_input_len,
                           min_output_len=args.min_output_len, max_output_len=args.max_output_len)
    mine = deal(trace, rank, world)
    longest = max((len(r.prompt_token_ids) + r.max_new_tokens for r in trace), default=1)
    slots = args.batch_size + max(1, args.staging_slots)
    pages_per_seq = (longest + args.page_size - 1) // args.page_size + 1
    kv_page_bytes = 2 * cfg["num_hidden_layers"] * cfg["num_key_value_heads"] * args.page_size * cfg["head_dim"] * 2
    clock = time.perf_counter
    if args.solution == "engine":
        import torch

        if not torch.cuda.is_available():
            raise SystemExit("serve_replicas needs one GPU per rank (or --solution schedule-only): the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        from tiny_llm_hip.engine import DecodeEngine
        from tiny_llm_hip.synthetic import synthetic_qwen3

        model = synthetic_qwen3(cfg, seed=args.seed, sigma=0.02, device=f"cuda:{local_rank}")
        engine = DecodeEngine(model, page_size=args.page_size, num_pages=pages_per_seq * slots + 2, max_batch=slots,
                              max_pages_per_seq=pages_per_seq,
                              max_prefill_rows=max(args.prefill_step, args.prefill_budget if args.staging_slots > 1 else 0, 8),
                              kv_format=args.kv_format)
    else:
        engine = ScheduleOnlyEngine(slots)
        clock = engine.clock

    def run(reqs):
        return serve_requests(engine, reqs, batch_size=args.batch_size, prefill_step=args.prefill_step,
                              prefill_budget=args.prefill_budget, page_size=args.page_size, kv_bytes_per_page=kv_page_bytes,
                              capacity_pages=pages_per_seq * slots + 2, clock=clock, staging_slots=args.staging_slots, compact=not args.keep_slot_holes)

    warm = args.warmup_requests if args.warmup_requests is not None else args.batch_size
    if warm > 0 and mine:
        run(mine[:warm])  # complete-request warm-up (graph captures for every row bucket the trace reaches)
    if dist is not None:
        dist.barrier()
    t0 = clock()
    metrics = run(mine)
    wall = clock() - t0
    report = {"rank": rank, "requests": len(mine), "request_indices": list(range(rank, len(trace), world)),  # deal(): request i -> replica i mod N
              "prompt_tokens": sum(len(r.prompt_token_ids) for r in mine), "wall_s": wall,
              "decode_step_ms": metrics.decode_step_ms, "metrics": {k: v for k, v in asdict(metrics).items() if k != "decode_step_ms"}}
    reports = [report]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, report)
        reports = gathered
        dist.barrier()
    out = None
    if rank == 0:
        out = aggregate(reports)
        print(f"Replicas: {out['replicas']} (requests dealt i mod N, {args.batch_size} decode slots each, no data-path collective)")
        print(f"Requests: {out['requests']}, Prompt tokens: {out['prompt_tokens']}, Generated tokens: {out['generated_tokens']}")
        print(f"Time: {out['wall_s']:.2f}s, Output throughput: {out['output_tok_s']:.2f} tok/s")
        print(f"Total throughput (prompt+output): {out['total_tok_s']:.2f} tok/s")
        print(f"Prefill throughput: {out['prefill_tok_s']:.2f} tok/s")
        print(f"Decode throughput: {out['decode_tok_s']:.2f} tok/s")
        print(f"Request throughput: {out['req_s']:.2f} req/s")
        print(f"Decode step p50/p95: {out['decode_step_p50_ms']:.3f} / {out['decode_step_p95_ms']:.3f} ms")
        for r in out["per_replica"]:
            print(f"  GPU {r['rank']}: {r['requests']} requests, {r['wall_s']:.2f}s, output {r['output_tok_s']:.2f} tok/s, "
                  f"decode {r['decode_tok_s']:.2f} tok/s, step p50/p95 {r['decode_step_p50_ms']:.3f}/{r['decode_step_p95_ms']:.3f} ms, "
                  f"peak active {r['peak_active_requests']}")
        if world == 1:  # the single-replica report in the reference's own words as well
            print("\n".join(report_lines(len(mine), report["prompt_tokens"], wall, metrics)[5:]))
        if args.json_output:
            args.json_output.parent.mkdir(parents=True, exist_ok=True)
            args.json_output.write_text(json.dumps({"config": {k: str(v) for k, v in vars(args).items()}, "aggregate": out}, indent=1))
        print(json.dumps(driver_line(args, out, world)), flush=True)  # ONE line in bench.py's shape: the last line of stdout
    if args.solution == "engine":
        engine.close()
    if dist is not None:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
