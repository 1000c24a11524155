"""CPU tier: the serving scheduler (benches/serving.py) and the config-4 multi-replica dealer (benches/serve_replicas.py)
against the schedule-only engine (a cost model; no kernel runs).  The world-size-2 case goes through
torch.distributed.run with the gloo backend, exactly as the GPU launch does with one process per GPU."""

import json
import subprocess
import sys
from pathlib import Path
from random import Random

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _trace(n, seed=0):
    from benches.bench import build_requests

    return build_requests(rng=Random(seed), num_seqs=n, vocab_size=151936, eos_token_id=151935, min_input_len=128,
                          max_input_len=1024, min_output_len=32, max_output_len=128)


def test_reference_schedule_starves_a_64_slot_batch_and_a_prefill_budget_fills_it():
    """One 128-token chunk per turn (the reference's schedule, batch.py:136-285) admits requests about as fast as they
    finish, so most of 64 slots stay empty (r01: 21 busy); 2048 prompt tokens of admission per turn fills all of them.
    Every request is served exactly once either way, with exactly its token budget."""
    from benches.serving import ScheduleOnlyEngine, serve_requests

    reqs = _trace(160)
    want_tokens = sum(r.max_new_tokens for r in reqs)
    results = {}
    for budget in (128, 2048):
        eng = ScheduleOnlyEngine(65)
        m = serve_requests(eng, reqs, batch_size=64, prefill_step=128, prefill_budget=budget, clock=eng.clock)
        assert m.generated_tokens == want_tokens and all(s is None for s in eng.slots)
        assert m.decode_tokens == want_tokens - len(reqs)  # the first token of a request comes out of its prefill
        assert m.prefill_chunks == sum(-(-len(r.prompt_token_ids) // 128) for r in reqs)
        results[budget] = (m.peak_active_requests, eng.now)
    assert results[128][0] < 40, results
    assert results[2048][0] >= 64, results
    assert results[2048][1] < results[128][1], "filling the batch must shorten the virtual makespan"


def test_report_lines_are_the_reference_strings():
    """The strings the reference's drivers regex (benches/bench_course_progression.py:103-105) and its serving report
    labels (benches/bench.py:787-830)."""
    import re

    from benches.serving import ScheduleOnlyEngine, report_lines, serve_requests

    eng = ScheduleOnlyEngine(5)
    reqs = _trace(6)
    m = serve_requests(eng, reqs, batch_size=4, prefill_step=128, clock=eng.clock)
    text = "\n".join(report_lines(len(reqs), sum(len(r.prompt_token_ids) for r in reqs), eng.now, m))
    for label in ("Prefill", "Decode", "Output"):
        assert re.search(rf"{label} throughput: ([0-9.]+) tok/s", text)
    for label in ("Request throughput:", "Peak active requests:", "Peak KV bytes:", "Peak live KV pages:",
                  "Peak KV capacity pages:", "Peak tail waste slots:", "Tail-waste snapshot live slots:",
                  "Tail-waste snapshot bytes:", "Tail-waste snapshot fraction:", "Decode step latency ms (median/p95/max):",
                  "Decode completion gap ms (median/p95/max):", "Reused page allocations:", "Page-pool growths:",
                  "Pages copied during pool growth:", "Dense KV bytes copied during growth:",
                  "Dense KV bytes copied into batch tensors:", "Paged KV bytes copied during pool growth:"):
        assert label in text, label
    assert m.peak_active_requests <= 5 and m.decode_step_count == len(m.decode_step_ms) > 0


def test_dealer_and_aggregate_are_consistent():
    from benches.serve_replicas import aggregate, deal

    reqs = list(range(10))
    assert deal(reqs, 0, 4) == [0, 4, 8] and deal(reqs, 3, 4) == [3, 7]
    assert sorted(sum((deal(reqs, r, 4) for r in range(4)), [])) == reqs
    reports = [{"rank": r, "requests": 2, "prompt_tokens": 100, "wall_s": 1.0 + r, "decode_step_ms": [1.0, 2.0 + r],
                "metrics": {"generated_tokens": 50, "decode_tokens": 48, "decode_time": 0.5, "prefill_time": 0.25,
                            "peak_active_requests": 2}} for r in range(2)]
    out = aggregate(reports)
    assert out["wall_s"] == 2.0 and out["output_tok_s"] == 100 / 2.0  # the job ends with its slowest replica
    assert out["decode_tok_s"] == 2 * 48 / 0.5 and out["req_s"] == 4 / 2.0
    assert out["decode_step_p95_ms"] == 3.0 and len(out["per_replica"]) == 2


def test_two_replicas_over_gloo_match_one_replica_serving_everything(tmp_path):
    """world size 2 through torch.distributed.run (gloo control plane): every request is served exactly once, the two
    replicas split the trace by index, and the virtual job time is about half of the single-replica run."""
    common = ["--solution", "schedule-only", "--num-seqs", "96", "--batch-size", "16", "--warmup-requests", "0"]
    one = tmp_path / "one.json"
    two = tmp_path / "two.json"
    subprocess.run([sys.executable, str(ROOT / "benches" / "serve_replicas.py"), *common, "--json-output", str(one)],
                   check=True, cwd=ROOT, timeout=300, capture_output=True)
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(ROOT / "benches" / "serve_replicas.py"),
                           *common, "--json-output", str(two)], cwd=ROOT, timeout=600, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert "Replicas: 2" in proc.stdout and "GPU 1:" in proc.stdout
    a, b = json.loads(one.read_text())["aggregate"], json.loads(two.read_text())["aggregate"]
    assert a["replicas"] == 1 and b["replicas"] == 2
    assert a["requests"] == b["requests"] == 96 and a["generated_tokens"] == b["generated_tokens"]
    assert [r["requests"] for r in b["per_replica"]] == [48, 48]
    assert 0.4 < b["wall_s"] / a["wall_s"] < 0.75
    # the driver-shaped record: ONE JSON line, the last of rank 0's stdout, in bench.py's shape with BASELINE.json configs[3] named
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and proc.stdout.strip().splitlines()[-1] == lines[0]
    line = json.loads(lines[0])
    assert line["metric"] == json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    assert line["n_gpus"] == 2 and line["unit"] == "tokens/s" and line["higher_is_better"] is True and line["scaling"] == "strong"
    assert "configs[3]" in line["config"]["workload"] and "continuous batching" in line["config"]["workload"]
    assert line["value"] == pytest.approx(b["output_tok_s"], rel=1e-3), "value is the whole-job aggregate over both replicas"
    assert [g["rank"] for g in line["per_gpu"]] == [0, 1] and all(g["output_tokens_per_s"] > 0 for g in line["per_gpu"])
    assert sum(g["output_tokens_per_s"] for g in line["per_gpu"]) >= line["value"] * 0.999  # the job ends with its slowest replica
    # bytes per decode step follow SURVEY 8d: the weights once + 147,456 B per live context token
    assert line["roofline"]["bytes_per_decode_step"] > 2_136_832_000 and len(line["roofline"]["achieved_per_gpu"]) == 2
    assert line["data"].startswith("none"), "a cost-model run must not read as a measurement"


def test_config1_week1_cpu_runner_smoke():
    """BASELINE config 1 (Qwen3-0.6B Week-1 greedy decode on the host CPU): the runner builds the dense Week-1 model of the
    host mirror on CPU tensors and generates greedily without a KV cache (shrunk to 2 layers / 4k vocabulary here)."""
    from benches.bench_config1 import main

    out = main(["--layers", "2", "--vocab", "4096", "--prompt-len", "8", "--new-tokens", "4"])
    assert out["generated_tokens"] == 4 and out["prompt_tokens"] == 8 and out["layers"] == 2
    assert all(0 <= t < 4096 for t in out["first_ids"]) and out["decode_tok_s"] > 0
    again = main(["--layers", "2", "--vocab", "4096", "--prompt-len", "8", "--new-tokens", "4"])
    assert again["first_ids"] == out["first_ids"], "seeded weights and prompt: the greedy ids must reproduce"


def test_packed_admission_schedule_only():
    """serve_requests with several staging slots (benches/serving.py _serve_requests_packed): every request gets exactly its
    tokens, a packed pass never exceeds the budget, 16 chunks or one chunk per slot, finished prompts enter the batch in
    admission order, all slots come back; with ONE staging slot the engine never sees a packed call (reference policy)."""
    from random import Random

    from benches.serving import ScheduleOnlyEngine, serve_requests

    class Req:
        def __init__(self, n, m):
            self.prompt_token_ids = list(range(1, n + 1))
            self.max_new_tokens = m

    rng = Random(5)
    reqs = [Req(rng.randint(1, 700), rng.randint(1, 40)) for _ in range(60)]

    class Spy(ScheduleOnlyEngine):
        def __init__(self, slots):
            super().__init__(slots)
            self.passes = []

        def prefill_packed(self, chunks):
            self.passes.append([(slot, len(toks), last) for slot, toks, last in chunks])
            super().prefill_packed(chunks)

    for staging, budget, step in ((4, 512, 512), (16, 2048, 128), (3, 100, 64)):
        eng = Spy(8 + staging)
        m = serve_requests(eng, reqs, batch_size=8, prefill_step=step, prefill_budget=budget, clock=eng.clock, staging_slots=staging)
        assert m.generated_tokens == sum(r.max_new_tokens for r in reqs)
        assert all(x is None for x in eng.slots), "every slot released"
        assert m.peak_active_requests <= 8 + staging
        assert eng.passes and max(len(p) for p in eng.passes) > 1, "prompts are really packed"
        for p in eng.passes:
            assert sum(n for _, n, _ in p) <= budget and len(p) <= 16 and all(n <= step for _, n, _ in p)
            assert all(slot >= 8 for slot, _, _ in p), "prefill happens in the staging slots only"
        assert sum(n for p in eng.passes for _, n, _ in p) == sum(len(r.prompt_token_ids) for r in reqs)
    eng = Spy(9)
    m = serve_requests(eng, reqs, batch_size=8, prefill_step=128, prefill_budget=2048, clock=eng.clock)
    assert not eng.passes and m.generated_tokens == sum(r.max_new_tokens for r in reqs)


def test_hole_closing_keeps_the_decode_prefix_dense():
    """benches/serving.py _close_holes: before a step the live decode slots are moved into the holes finished requests left whenever
    that lowers the row bucket -- every step then runs at the bucket of the NUMBER of live requests, each request still gets exactly
    its tokens and every slot comes back; the counters the reference's report prints do not see slot numbers (same values with the
    moves switched off), only the clock does."""
    from benches.serving import ROW_BUCKETS, ScheduleOnlyEngine, serve_requests

    class Spy(ScheduleOnlyEngine):
        def __init__(self, slots):
            super().__init__(slots)
            self.steps, self.moves = [], 0

        def move(self, src, dst):
            self.moves += 1
            super().move(src, dst)

        def decode(self, steps, batch=None):
            live = [i for i, c in enumerate(self.slots[:64]) if c is not None]
            self.steps.append((batch, len(live), max(live) + 1))
            assert all(c is None for c in self.slots[batch:64]), "a live request outside the decoded prefix"
            super().decode(steps, batch=batch)

    reqs = _trace(160)
    out = {}
    for compact, staging in ((True, 1), (False, 1), (True, 8), (False, 8)):
        eng = Spy(64 + staging)
        m = serve_requests(eng, reqs, batch_size=64, prefill_step=128, prefill_budget=128 if staging == 1 else 1024, clock=eng.clock,
                           staging_slots=staging, compact=compact)
        assert m.generated_tokens == sum(r.max_new_tokens for r in reqs) and all(s is None for s in eng.slots)
        bucket = lambda n: next(b for b in ROW_BUCKETS if b >= n)
        if compact:
            assert all(batch == bucket(count) for batch, count, _top in eng.steps), "a step ran at more rows than its live requests need"
        out[(compact, staging)] = (m, eng.now, eng.moves - len(reqs), sum(b for b, _, _ in eng.steps))
    for staging in (1, 8):
        on, off = out[(True, staging)], out[(False, staging)]
        assert on[2] > 0 and off[2] <= 0, "moves beyond the one admission move per request happen only when closing holes"
        assert on[3] < off[3] and on[1] < off[1], "fewer decoded rows, shorter virtual makespan"
    # the reference's schedule (one staging slot): the admission order does not depend on slot numbers, so every reported counter is equal
    a, b = out[(True, 1)][0], out[(False, 1)][0]
    for name in ("turns", "prefill_chunks", "decode_tokens", "generated_tokens", "peak_active_requests", "peak_live_pages", "peak_tail_waste_slots",
                 "decode_step_count", "decode_bytes"):
        assert getattr(a, name) == getattr(b, name), name


def test_profile_week2_kernels_harness_logic(monkeypatch):
    """benches/profile_week2_kernels.py (reference: benches/profile_week2_kernels.py:88-156): the case syntax, the rotated
    group order and the median -- against a fake clock, no GPU."""
    import argparse

    from benches import profile_week2_kernels as P

    assert P.parse_case("split-k:prefill:32") == P.ProfileCase("split-k", "prefill", 32)
    for bad in ("split-k:prefill", "a:warmup:3", "a:decode:0", "a:decode:x"):
        with pytest.raises(argparse.ArgumentTypeError):
            P.parse_case(bad)
    assert [c.checkpoint for c in P.parse_args([]).case] == [c.split(":")[0] for c in P.DEFAULT_CASES]
    calls, now = [], [0.0]
    cost = {"a": 1.0, "b": 2.0, "c": 4.0}

    def builder(name):
        def build():
            calls.append(name)
            return [name]
        return build

    def evaluate(outputs):
        now[0] += cost[outputs[0]]

    monkeypatch.setattr(P, "perf_counter", lambda: now[0])
    got = P.benchmark_groups([(n, builder(n)) for n in "abc"], warmup=2, iterations=3, evaluate=evaluate)
    assert got == {"a": 1e6, "b": 2e6, "c": 4e6}
    assert "".join(calls) == "abc" "bca" "cab" "abc" "bca", "order rotated by one every round, warm-up rounds included"


def test_serve_replicas_starts_its_own_ranks_when_no_launcher_did():
    """`python benches/serve_replicas.py --gpus 2 ...` without torch.distributed.run around it: the script launches its two ranks
    itself (gloo control plane, schedule-only engine here) and rank 0 reports two replicas."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = ROOT / "gpurun_out"
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        report = Path(tmp) / "report.json"
        proc = subprocess.run([sys.executable, str(ROOT / "benches" / "serve_replicas.py"), "--gpus", "2", "--solution", "schedule-only",
                               "--num-seqs", "24", "--batch-size", "8", "--json-output", str(report)], env=env, capture_output=True,
                              text=True, timeout=300)
        assert proc.returncode == 0, proc.stderr[-2000:]
        data = json.loads(report.read_text())
    text = json.dumps(data)
    assert '"replicas": 2' in text, text[:500]


def test_eight_ranks_rehearsal_of_both_launch_paths():
    """The shape the driver runs on an 8-GPU node (SCALE_rNN: `python bench.py --gpus 8 ...`; config 4: `serve_replicas.py --gpus 8`),
    rehearsed at world size 8 over gloo without a GPU: rendezvous on 127.0.0.1, barrier, MAX over ranks, one line from rank 0 --
    `per_rank` holds eight clocks, and request i of the seeded trace is served by replica i mod 8 (requests shard,
    nothing else does; no collective on the data path)."""
    import os
    import tempfile

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--engine", "sleep"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["parallelism"].startswith("request-parallel x8")
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and len(pr["tokens_per_s"]) == 8 and 0 <= pr["slowest_rank"] < 8
    assert out["value"] == pytest.approx(8 * 4 / (out["ms_per_step"] * 4 / 1e3), rel=1e-3), "whole-job aggregate over the eight ranks"

    with tempfile.TemporaryDirectory() as tmp:
        report = Path(tmp) / "report.json"
        proc = subprocess.run([sys.executable, str(ROOT / "benches" / "serve_replicas.py"), "--gpus", "8", "--solution", "schedule-only",
                               "--num-seqs", "43", "--batch-size", "4", "--json-output", str(report)], env=env, capture_output=True,
                              text=True, timeout=600)
        assert proc.returncode == 0, proc.stderr[-2000:]
        data = json.loads(report.read_text())
    agg = data["aggregate"]
    assert agg["replicas"] == 8 and agg["requests"] == 43
    assert [r["rank"] for r in agg["per_replica"]] == list(range(8))
    for r in agg["per_replica"]:
        assert r["request_indices"] == list(range(r["rank"], 43, 8)), "request i -> replica i mod 8"
        assert r["requests"] == len(r["request_indices"])
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and json.loads(line[0])["n_gpus"] == 8 and len(json.loads(line[0])["per_gpu"]) == 8
