"""GPU tier: the two replay routes of a captured decode step must be THE SAME COMPUTATION.

Default: AQL dispatch packets on the engine's own HSA queue, write-through stores + per-layer hand-over buffers instead of cache
maintenance between the launches (csrc/aql.h, common.h `act_store`, csrc/engine.hip per-layer decode activations).  TL_AQL=0: hipGraphLaunch
of the same captured step (plain stores, agent-scope fences around every launch).  Same kernels' arithmetic, same order: greedy ids AND
logits must be bit-identical -- at every attention plan a single sequence goes through (1 / 2 / 4 / 8 windows merged by the wo GEMV,
16+ windows with the merge launch, the GQA-group walk on the matrix cores), at 2 and 4 sequences (fused GEMV with 2 / 4 rows), across
a page boundary inside one call (a stream-side poke between two steps), and behind a profiled (eager) step.  5 .. 64 sequences (since round 5 on
the route too: the register-resident and K-sliced matmuls with write-through stores, per-layer rows / weighted rows / slice planes) at every
routing boundary of the batched step (8 | 9 rows: fragment order; 16 | 17 and 32 | 33: the wo projection changes kernels); a plan whose
attention windows do not fit the per-layer partials must stay on the graph route, silently and correctly.
Reference loop both routes implement: src/tiny_llm_ref/qwen3_week3.py:55-121,320-338."""

import os

import numpy as np
import pytest
import torch

from helpers import QWEN4B_CFG

pytestmark = pytest.mark.gpu

CFG = dict(QWEN4B_CFG, num_hidden_layers=3, vocab_size=32768)


@pytest.fixture(scope="module")
def model():
    from tiny_llm_hip.synthetic import synthetic_qwen3

    return synthetic_qwen3(CFG, seed=17, sigma=0.02, device="cuda")


def _run(model, route, prompts, steps, page=128, chunk=2048, profile_between=False, calls=1):
    from tiny_llm_hip.engine import DecodeEngine

    old = os.environ.pop("TL_AQL", None)
    if route == "hipgraph":
        os.environ["TL_AQL"] = "0"
    try:
        n = len(prompts)
        pages = sum((len(p) + steps * calls + 2 * page) // page + 1 for p in prompts) + 2
        eng = DecodeEngine(model, page_size=page, num_pages=pages, max_batch=n, max_prefill_rows=max(8, min(2048, max(len(p) for p in prompts))))
        try:
            assert eng.replay_route().startswith(route), eng.replay_route()
            for i, p in enumerate(prompts):
                eng.begin(i)
                eng.prefill(i, p, chunk=chunk)
            eng.decode(3, batch=n)  # eager warm step, capture, first replays
            for _ in range(calls):
                if profile_between:
                    eng.profile_step(n)  # an eager, stamped step on the stream between two replayed runs
                eng.decode(steps, batch=n)
            ids = [eng.read_tokens(i, 3 + steps * calls + (calls if profile_between else 0)) for i in range(n)]
            logits = eng.logits(n).float().cpu().numpy()
            st = eng.stats()
            for i in range(n):
                eng.release(i)
            return ids, logits, st
        finally:
            eng.close()
    finally:
        os.environ.pop("TL_AQL", None)
        if old is not None:
            os.environ["TL_AQL"] = old


def _prompts(lengths, seed):
    rng = np.random.default_rng(seed)
    return [[int(t) for t in rng.integers(16, CFG["vocab_size"], size=n)] for n in lengths]


@pytest.mark.parametrize("lengths", [[40], [100], [130], [300], [1500], [3000], [5000], [200, 90], [300, 17, 120, 64]])
def test_aql_route_is_bit_identical_to_the_graph_route(model, lengths):
    prompts = _prompts(lengths, sum(lengths))
    steps = 12
    ids_a, log_a, st_a = _run(model, "aql", prompts, steps)
    ids_g, log_g, st_g = _run(model, "hipgraph", prompts, steps)
    assert st_a["aql_steps"] >= steps and st_g["aql_steps"] == 0, (st_a, st_g)
    assert st_a["graph_replays"] == st_g["graph_replays"]
    assert ids_a == ids_g, f"greedy ids differ between the routes: {lengths}"
    assert np.array_equal(log_a, log_g), f"logits differ between the routes by up to {np.abs(log_a - log_g).max()}: {lengths}"


def test_a_page_boundary_and_a_profiled_step_inside_a_replayed_run(model):
    """16-token pages: every 16 steps a sequence crosses into a new page -- a stream-side poke of the block table between two steps of one
    call (the queue is drained, the poke runs, the run goes on); and an eager profiled step between two calls."""
    prompts = _prompts([37, 21], 5)
    ids_a, log_a, st_a = _run(model, "aql", prompts, steps=40, page=16, chunk=64, profile_between=True, calls=2)
    ids_g, log_g, st_g = _run(model, "hipgraph", prompts, steps=40, page=16, chunk=64, profile_between=True, calls=2)
    assert st_a["aql_steps"] >= 70 and ids_a == ids_g and np.array_equal(log_a, log_g)


@pytest.mark.parametrize("n_seqs", [5, 8, 9, 16, 17, 32, 33, 64])
def test_batched_steps_are_bit_identical_on_both_routes(model, n_seqs):
    """The batched-matmul step (csrc/qmm6.h, qmm3.h) on the AQL route: same ids and logits as hipGraphLaunch of the same captured step."""
    rng = np.random.default_rng(n_seqs)
    lengths = [int(x) for x in rng.integers(20, 300, size=n_seqs)]
    prompts = _prompts(lengths, n_seqs)
    ids_a, log_a, st_a = _run(model, "aql", prompts, steps=10)
    ids_g, log_g, st_g = _run(model, "hipgraph", prompts, steps=10)
    assert st_a["aql_steps"] >= 10 and st_g["aql_steps"] == 0, (st_a, st_g)
    assert ids_a == ids_g, f"greedy ids differ between the routes at {n_seqs} sequences"
    assert np.array_equal(log_a, log_g), f"logits differ between the routes by up to {np.abs(log_a - log_g).max()} at {n_seqs} sequences"


def test_batched_steps_over_many_steps_and_page_boundaries(model):
    """12 sequences, 16-token pages, 60 steps in two calls with a profiled step between them: pokes between steps, the window plan changes."""
    prompts = _prompts([37, 21, 50, 64, 90, 10, 33, 47, 120, 15, 70, 28], 12)
    ids_a, log_a, st_a = _run(model, "aql", prompts, steps=30, page=16, chunk=64, profile_between=True, calls=2)
    ids_g, log_g, st_g = _run(model, "hipgraph", prompts, steps=30, page=16, chunk=64, profile_between=True, calls=2)
    assert st_a["aql_steps"] >= 40 and ids_a == ids_g and np.array_equal(log_a, log_g)


def test_a_plan_beyond_the_per_layer_partials_stays_on_the_graph_route(model):
    """8 sequences behind 3,000-token prompts: 8 x 16 windows x 32 heads of partials exceed nothing yet, 64 sequences x 16 windows would -- the engine
    decides per plan; whatever it decides, the results equal the graph route's."""
    prompts = _prompts([3000] * 8, 88)
    ids_a, log_a, st_a = _run(model, "aql", prompts, steps=6)
    ids_g, log_g, st_g = _run(model, "hipgraph", prompts, steps=6)
    assert ids_a == ids_g and np.array_equal(log_a, log_g)


def test_tl_aql_1_is_accepted_and_0_is_the_graph_route(model, monkeypatch):
    from tiny_llm_hip.engine import DecodeEngine

    monkeypatch.setenv("TL_AQL", "1")
    eng = DecodeEngine(model, page_size=128, num_pages=4, max_batch=1, max_prefill_rows=8)
    assert eng.replay_route() == "aql"
    eng.close()
    monkeypatch.setenv("TL_AQL", "0")
    eng = DecodeEngine(model, page_size=128, num_pages=4, max_batch=1, max_prefill_rows=8)
    assert eng.replay_route().startswith("hipgraph")
    eng.close()
