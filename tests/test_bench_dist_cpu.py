"""CPU tier, world_size 2 over gloo: the request-parallel launch path of bench.py — per-rank request shards with no
data-path collective, barrier-bracketed timing, MAX over ranks, whole-job aggregate (driver contract)."""

import os
import random
import socket
import sys
import time
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, steps, queue):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world)
    prompt = bench.build_prompt(random.Random(0 * 1000 + rank), 16, 1000)  # same derivation as bench.main
    step_time = 0.002 * (rank + 1)  # rank 1 is the slow rank

    def run(k):
        time.sleep(step_time * k)

    elapsed_max, local = bench.timed_steps(run, lambda: None, steps, dist, torch.device("cpu"))
    queue.put((rank, prompt, elapsed_max, local, bench.aggregate_value(world, steps, elapsed_max)))
    dist.destroy_process_group()


def test_two_rank_request_parallel_timing():
    world, steps = 2, 25
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(queue.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, prompt0, max0, local0, value0), (r1, prompt1, max1, local1, value1) = got
    assert prompt0 != prompt1, "every rank must decode its own request"
    assert max0 == pytest.approx(max1, rel=1e-9), "all ranks agree on the MAX-reduced elapsed time"
    assert max0 >= local1 * 0.999 and local1 > local0, "the job takes as long as its slowest rank"
    assert value0 == pytest.approx(world * steps / max0)
    assert value0 < world * steps / local0, "aggregate must not be extrapolated from the fast rank"


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, str(ROOT))
    import bench

    elapsed, local = bench.timed_steps(lambda k: time.sleep(0.001 * k), lambda: None, 10)
    assert elapsed == local and elapsed >= 0.01
    assert bench.aggregate_value(1, 10, elapsed) == pytest.approx(10 / elapsed)


def test_bench_py_starts_its_own_ranks_when_no_launcher_did():
    """The driver's command shape is `python3 bench.py --gpus N ...` (BENCH_rNN.json "cmd").  With N > 1 and WORLD_SIZE unset the
    script starts N ranks of itself under torch.distributed.run on 127.0.0.1; rank 0 prints the one JSON line.  Exercised here at
    world size 2 over gloo with --engine sleep (the launch / rendezvous / barrier / MAX-over-ranks / report path without a GPU)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--engine", "sleep"],
                          env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] == pytest.approx(2 * 6 / (out["ms_per_step"] * 6 / 1e3), rel=1e-3)  # whole-job aggregate over both ranks
    assert out["data"].startswith("none") and out["config"]["parallelism"].startswith("request-parallel x2")
    # every rank's own clock, gathered once after the timed region: a straggler GPU must be visible in the driver's record
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["tokens_per_s"]) == 2 and pr["slowest_rank"] in (0, 1)
    assert pr["min_tokens_per_s"] == min(pr["tokens_per_s"]) and pr["max_tokens_per_s"] == max(pr["tokens_per_s"])
    assert max(pr["ms_per_step"]) == pytest.approx(out["ms_per_step"], rel=0.05), "the job's step time is its slowest rank's"
