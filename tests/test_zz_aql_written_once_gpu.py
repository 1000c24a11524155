"""GPU tier: the AQL replay route's invariant as a CHECKED property, and a soak of the two routes against each other.

The route replays a captured decode step with no cache maintenance between its launches (csrc/aql.h).  That is correct because every
address one launch hands to a later launch of the step is written ONCE per step and read only after it (write-through stores, per-layer
hand-over buffers: csrc/engine.hip, common.h `act_store`).  Round 5 found a violation by review -- weighted-row buffers sized by slot
count, a 9-row batch wrote a block tail into the neighbouring buffer -- after the bit-identity tests had been green on it: identity over
a few dozen steps samples the invariant, it does not enforce it.  Here:

  * tl_engine_check_step (include/tinyllm_engine.h) runs one real decode step with the per-layer buffers POISONED and a checker behind
    every launch that diffs the hand-over regions against a shadow copy, per 2-byte element: a second write of the step to an element is
    counted and located; a value read before it was written reaches the logits as NaN.  Run over the plans a single sequence goes through
    (1 / 2 / 4 / 8 / 16 / 32 windows; one query head or a GQA group per workgroup; both attention walks), 2-4 sequences, and 5 .. 64
    sequences at every routing boundary of the batched step -- the plans tests/test_decode_plans_cpu.py enumerates.
  * the soak: thousands of consecutive decode steps at 1 / 9 / 33 / 64 sequences with page crossings (16-token pages: a poke every 16
    steps per sequence), finished sequences released, holes closed by tl_engine_move, new sequences forked from live ones, the window
    plan changing as contexts grow -- AQL route against hipGraphLaunch of the same captured steps: every greedy id and a checksum of
    every step's logits equal.
Reference loop both routes implement: src/tiny_llm_ref/qwen3_week3.py:55-121,320-338; batch.py:136-285 for the slot churn."""

import os
import zlib

import numpy as np
import pytest
import torch

from helpers import QWEN4B_CFG

pytestmark = pytest.mark.gpu

CFG = dict(QWEN4B_CFG, num_hidden_layers=3, vocab_size=32768)


@pytest.fixture(scope="module")
def model():
    from tiny_llm_hip.synthetic import synthetic_qwen3

    return synthetic_qwen3(CFG, seed=23, sigma=0.02, device="cuda")


def _prompts(lengths, seed):
    rng = np.random.default_rng(seed)
    return [[int(t) for t in rng.integers(16, CFG["vocab_size"], size=n)] for n in lengths]


def _engine(model, prompts, page=128, extra_tokens=64, max_batch=None, chunk=2048):
    from tiny_llm_hip.engine import DecodeEngine

    n = len(prompts)
    pages = sum((len(p) + extra_tokens + 2 * page) // page + 1 for p in prompts) + 2
    eng = DecodeEngine(model, page_size=page, num_pages=pages, max_batch=max_batch or n, max_prefill_rows=max(8, min(2048, max(len(p) for p in prompts))))
    for i, p in enumerate(prompts):
        eng.begin(i)
        eng.prefill(i, p, chunk=chunk)
    return eng


SINGLE = [[40], [100], [130], [300], [700], [1500], [3000], [5000], [200, 90], [300, 17, 120, 64], [1500, 1400]]
BATCHED = [5, 8, 9, 16, 17, 24, 32, 33, 48, 49, 64]


def _check_plan(model, lengths, label):
    """A checked step between replayed steps: no element written twice, nothing read before it was written (finite logits, equal to the
    logits of an engine that took the same steps unchecked), and the plan is one the route replays."""
    prompts = _prompts(lengths, 7 + sum(lengths))
    n = len(prompts)
    eng, twin = _engine(model, prompts), _engine(model, prompts)
    try:
        assert eng.replay_route() == "aql"
        eng.decode(3, batch=n)
        twin.decode(3, batch=n)
        for _ in range(2):
            rep = eng.check_step(n)
            twin.decode(1, batch=n)
            assert rep["launches"] > 0 and rep["elements_written"] > 0, rep
            assert rep["double_writes"] == 0, (f"{label}: {rep['double_writes']} elements written twice in one step; first by launch {rep['first_launch']} "
                                               f"(kind {rep['first_kind']}), region {rep['first_region']}, element {rep['first_offset']}")
            assert rep["written_once_plan"] == 1, f"{label}: the plan is not replayed on the AQL route ({rep})"
            got, want = eng.logits(n).float(), twin.logits(n).float()
            assert torch.isfinite(got).all(), f"{label}: {int((~torch.isfinite(got)).sum())} logits are not finite: a poisoned element was read before the step wrote it"
            assert torch.equal(got, want), f"{label}: the checked step's logits differ from the replayed step's by up to {float((got - want).abs().max())}"
            eng.decode(2, batch=n)
            twin.decode(2, batch=n)
        ids = [eng.read_tokens(i, 9) for i in range(n)]
        assert ids == [twin.read_tokens(i, 9) for i in range(n)], f"{label}: greedy ids differ behind a checked step"
    finally:
        eng.close()
        twin.close()


@pytest.mark.parametrize("lengths", SINGLE, ids=lambda l: "x".join(map(str, l)))
def test_every_hand_over_address_is_written_once_per_step(model, lengths):
    _check_plan(model, lengths, f"contexts {lengths}")


@pytest.mark.parametrize("n_seqs", BATCHED)
def test_the_batched_step_writes_every_hand_over_address_once(model, n_seqs):
    rng = np.random.default_rng(n_seqs)
    _check_plan(model, [int(x) for x in rng.integers(20, 300, size=n_seqs)], f"{n_seqs} sequences")


def test_the_checker_sees_a_second_write(model):
    """The checker itself: a plan that keeps the SHARED buffers (TL_AQL=0 engines have no per-layer ones) rewrites x, h, the qkv rows ...
    in every layer -- the checker must count those second writes, name the launch, and say that the plan is not a written-once one."""
    from tiny_llm_hip.engine import DecodeEngine

    os.environ["TL_AQL"] = "0"
    try:
        prompts = _prompts([50], 3)
        eng = _engine(model, prompts)
        eng.decode(3, batch=1)
        rep = eng.check_step(1)
        eng.close()
    finally:
        os.environ.pop("TL_AQL", None)
    assert rep["written_once_plan"] == 0 and rep["double_writes"] > 1000 and rep["first_region"] == 0 and rep["first_launch"] >= 1, rep


# ---- soak ------------------------------------------------------------------------------------------------------------------------

def _soak(model, route, n_seqs, total_steps, seed, page=16):
    """A serving-like run: `n_seqs` slots on 16-token pages; every call decodes a few steps, then some sequences finish (released), the
    holes are closed (tl_engine_move of the last live slot), new ones arrive (prefilled, or forked from a live one).  Returns every id and
    a CRC of every call's logits."""
    from tiny_llm_hip.engine import DecodeEngine

    old = os.environ.pop("TL_AQL", None)
    if route == "hipgraph":
        os.environ["TL_AQL"] = "0"
    try:
        rng = np.random.default_rng(seed)
        cap = 704  # tokens a sequence may reach (prompt + forked prefix + steps)
        per_seq = cap // page + 2
        eng = DecodeEngine(model, page_size=page, num_pages=n_seqs * per_seq + 64, max_batch=n_seqs, max_pages_per_seq=per_seq, max_prefill_rows=64)
        assert eng.replay_route().startswith(route)
        live = 0                      # slots [0, live) are occupied
        budget = []                   # steps left per slot
        trace, crc = [], 0

        def admit(fork_from=None):
            nonlocal live
            slot = live
            if fork_from is not None:
                eng.fork(fork_from, slot)
            else:
                eng.begin(slot)
                eng.prefill(slot, [int(t) for t in rng.integers(16, CFG["vocab_size"], size=int(rng.integers(3, 60)))], chunk=64)
            budget.append(max(1, min(int(rng.integers(20, 400)), cap - 8 - eng.context_len(slot))))
            live += 1

        for _ in range(n_seqs):
            admit()
        done = 0
        while done < total_steps:
            k = int(rng.integers(1, 24))
            k = min(k, total_steps - done, min(budget))
            eng.decode(k, batch=live)
            done += k
            for s in range(live):
                budget[s] -= k
                trace.append(eng.read_tokens(s, k))
            crc = zlib.crc32(eng.logits(live).view(torch.int16).cpu().numpy().tobytes(), crc)
            # finished sequences leave; the last live slot moves into each hole (benches/serving.py _close_holes)
            s = 0
            while s < live:
                if budget[s] > 0:
                    s += 1
                    continue
                eng.release(s)
                last = live - 1
                if s != last:
                    eng.move(last, s)
                    budget[s] = budget[last]
                budget.pop()
                live -= 1
            while live < n_seqs:  # arrivals: a fresh prompt, or (one in three) a fork of a live sequence
                admit(fork_from=int(rng.integers(0, live)) if live > 0 and rng.integers(0, 3) == 0 else None)
        st = eng.stats()
        eng.close()
        return trace, crc, st
    finally:
        os.environ.pop("TL_AQL", None)
        if old is not None:
            os.environ["TL_AQL"] = old


@pytest.mark.parametrize("n_seqs,total_steps,page", [(1, 4096, 16), (9, 4096, 16), (33, 4096, 64), (64, 4096, 16)])
def test_soak_of_the_two_routes(model, n_seqs, total_steps, page):
    """(64-token pages at 33 sequences: the GQA-group walk on the matrix cores needs pages of 32 tokens or more; 16-token pages elsewhere: a
    block-table poke between two steps of a call nearly every step at 64 sequences)"""
    trace_a, crc_a, st_a = _soak(model, "aql", n_seqs, total_steps, seed=100 + n_seqs, page=page)
    trace_g, crc_g, st_g = _soak(model, "hipgraph", n_seqs, total_steps, seed=100 + n_seqs, page=page)
    assert st_a["decode_steps"] == st_g["decode_steps"] >= total_steps
    assert st_a["aql_steps"] >= 0.9 * total_steps and st_g["aql_steps"] == 0, (st_a["aql_steps"], total_steps)
    assert len(trace_a) == len(trace_g)
    bad = [i for i, (a, g) in enumerate(zip(trace_a, trace_g)) if a != g]
    assert not bad, f"{n_seqs} sequences: greedy ids differ between the routes first at read {bad[0]} of {len(trace_a)}"
    assert crc_a == crc_g, f"{n_seqs} sequences: the logits checksums of the two routes differ"
