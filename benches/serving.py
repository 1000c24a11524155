"""Continuous-batching serving loop over the decode engine, with the reference harness's report
(reference: benches/bench.py:35-62 ServingMetrics, :351-572 run_batch_requests_serving, :787-830 report lines).

One loop turn = (a) prefill work for the request being admitted, (b) hand it to a free decode slot once its prompt is
in the cache, (c) ONE batched decode step over the occupied slots, (d) retire finished requests.  The reference does
exactly one prefill chunk per turn; on an MI355X that makes admission the bottleneck of a 64-slot batch (a decode
step takes ~4 ms, a 128-token chunk ~7 ms, and requests finish as fast as they are admitted: 21 of 64 slots busy).
``prefill_budget`` therefore lets a turn spend up to that many prompt tokens on admission (default = one chunk, the
reference's schedule), which is what lets config 4 ("64 concurrent requests") actually reach 64.

The engine is duck-typed (begin / prefill / move / decode / release / synchronize / stats): ``ScheduleOnlyEngine``
runs the same schedule against a cost model instead of a GPU (capacity planning, and the CPU tests of the scheduler
and of the multi-replica dealer).
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field

# decode row counts the engine keeps captured graphs for (tiny_llm_hip.engine._DECODE_ROW_BUCKETS)
ROW_BUCKETS = (1, 2, 3, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256)


@dataclass
class ServingMetrics:
    """Field names are the reference's (benches/bench.py:35-62): drivers read them from the JSON payload.  Pages here are
    ENGINE pages (one id spans every layer's K and V: kv_bytes_per_page bytes); the growth / copy counters are zero by
    construction -- the pools are sized once, nothing is ever copied to grow (288 GB of HBM)."""
    generated_tokens: int = 0
    decode_tokens: int = 0
    prefill_time: float = 0.0
    decode_time: float = 0.0
    peak_active_requests: int = 0
    peak_live_pages: int = 0
    peak_capacity_pages: int = 0
    peak_tail_waste_slots: int = 0
    peak_tail_waste_live_slots: int = 0
    peak_tail_waste_bytes: int = 0
    peak_tail_waste_fraction: float = 0.0
    peak_kv_bytes: int = 0
    decode_step_count: int = 0
    decode_step_median_ms: float = 0.0
    decode_step_p95_ms: float = 0.0
    decode_step_max_ms: float = 0.0
    decode_gap_count: int = 0
    decode_gap_median_ms: float = 0.0
    decode_gap_p95_ms: float = 0.0
    decode_gap_max_ms: float = 0.0
    reused_page_allocations: int = 0
    storage_growths: int = 0
    copied_pages_on_growth: int = 0
    paged_growth_copy_bytes: int = 0
    dense_growth_copy_bytes: int = 0
    dense_staging_copy_bytes: int = 0
    # not in the reference: what the scheduler did
    prefill_chunks: int = 0
    turns: int = 0
    decode_bytes: int = 0  # algorithmic HBM bytes of all decode steps: per step W + 147,456 B x sum of live contexts
    decode_step_ms: list = field(default_factory=list, repr=False)


def nearest_rank(values, q: float) -> float:
    """Nearest-rank percentile, the reference's definition (benches/bench.py:579-585)."""
    if not values:
        return 0.0
    ordered = sorted(values)
    rank = max(1, min(len(ordered), -int(-q * len(ordered) // 1)))
    return ordered[rank - 1]


def median(values) -> float:
    if not values:
        return 0.0
    ordered = sorted(values)
    mid = len(ordered) // 2
    return ordered[mid] if len(ordered) % 2 else 0.5 * (ordered[mid - 1] + ordered[mid])


def _row_bucket(n: int, batch_size: int) -> int:
    return min(next((b for b in ROW_BUCKETS if b >= n), batch_size), batch_size)


def _close_holes(engine, slots: list, live: set) -> None:
    """Move the highest live decode slots into the holes finished requests left, when that lowers the step's row bucket.

    The engine decodes the occupied PREFIX of the slots and a step's cost follows its row count (a staircase in 16-row blocks:
    17-32 rows 1.77 ms, 33-64 rows 2.5 ms on the Qwen3-4B shape); with holes, 27 live requests spread over 48 slots pay for 48
    rows.  A move hands over a block-table row, a context length and the pending token (tl_engine_move: no K/V byte moves).  The
    reference decodes all ``batch_size`` rows of its batch cache every step (batch.py:136-285): slot numbers are invisible to it
    and to every counter of the report."""
    count = sum(s is not None for s in slots)
    top = max((i for i, s in enumerate(slots) if s is not None), default=-1)
    if count == 0 or _row_bucket(count, len(slots)) >= _row_bucket(top + 1, len(slots)):
        return
    lo, hi = 0, len(slots) - 1
    while True:
        while lo < hi and slots[lo] is not None:
            lo += 1
        while hi > lo and slots[hi] is None:
            hi -= 1
        if lo >= hi:
            return
        engine.move(hi, lo)
        slots[lo], slots[hi] = slots[hi], None
        live.discard(hi)
        live.add(lo)


def serve_requests(engine, requests, *, batch_size: int, prefill_step: int, prefill_budget: int | None = None,
                   page_size: int = 128, kv_bytes_per_page: int = 0, capacity_pages: int = 0,
                   clock=time.perf_counter, staging_slots: int = 1, compact: bool = True) -> ServingMetrics:
    """Serve ``requests`` (objects with prompt_token_ids, max_new_tokens) through ``engine`` with ``batch_size`` decode slots
    and ``staging_slots`` staging slots (indices batch_size ..).  Timers wrap a synchronised engine, like the reference's
    mx.eval inside them.  One staging slot = the reference's policy (one request prefilled at a time, batch.py:48-76);
    several = the admitted prompts' chunks go through ONE packed multi-token pass per turn (engine.prefill_packed), at most
    ``prefill_budget`` rows together.  ``compact``: close the holes in the decode slots before a step whenever that lowers its row
    bucket (_close_holes)."""
    if prefill_budget is None:
        prefill_budget = prefill_step
    if staging_slots > 1:
        return _serve_requests_packed(engine, requests, batch_size=batch_size, prefill_step=prefill_step,
                                      prefill_budget=prefill_budget, page_size=page_size, kv_bytes_per_page=kv_bytes_per_page,
                                      capacity_pages=capacity_pages, clock=clock, staging_slots=staging_slots, compact=compact)
    m = ServingMetrics()
    staging = batch_size
    slots: list[dict | None] = [None] * batch_size
    pending: dict | None = None
    next_idx = 0
    live: set[int] = set()
    gaps_ms: list[float] = []
    last_completion: float | None = None

    def snapshot():
        states = [s for s in slots if s is not None] + ([pending] if pending is not None else [])
        m.peak_active_requests = max(m.peak_active_requests, len(states))
        pages = sum((s["ctx"] + page_size - 1) // page_size for s in states)
        waste = sum((-s["ctx"]) % page_size for s in states if s["ctx"] > 0)
        m.peak_live_pages = max(m.peak_live_pages, pages)
        m.peak_capacity_pages = max(m.peak_capacity_pages, capacity_pages)
        m.peak_kv_bytes = max(m.peak_kv_bytes, capacity_pages * kv_bytes_per_page)
        if waste > m.peak_tail_waste_slots:
            m.peak_tail_waste_slots = waste
            m.peak_tail_waste_live_slots = pages * page_size
            m.peak_tail_waste_bytes = waste * (kv_bytes_per_page // page_size if page_size else 0)
            m.peak_tail_waste_fraction = waste / (pages * page_size) if pages else 0.0

    try:
        while next_idx < len(requests) or pending is not None or any(s is not None for s in slots):
            m.turns += 1
            budget = prefill_budget
            while budget > 0:
                if pending is None:
                    if next_idx >= len(requests):
                        break
                    engine.begin(staging)
                    live.add(staging)
                    pending = {"req": requests[next_idx], "offset": 0, "count": 0, "ctx": 0}
                    next_idx += 1
                tokens = pending["req"].prompt_token_ids
                if pending["offset"] < len(tokens):
                    chunk = tokens[pending["offset"]:pending["offset"] + prefill_step]
                    last = pending["offset"] + len(chunk) >= len(tokens)
                    t0 = clock()
                    engine.prefill(staging, chunk, chunk=len(chunk), want_logits=last)
                    engine.synchronize()
                    m.prefill_time += clock() - t0
                    m.prefill_chunks += 1
                    budget -= len(chunk)
                    pending["offset"] += len(chunk)
                    pending["ctx"] += len(chunk)
                    if last:
                        pending["count"] = 1
                        m.generated_tokens += 1
                    snapshot()
                if pending["offset"] < len(tokens):
                    continue
                if pending["count"] >= pending["req"].max_new_tokens:  # a one-token request never enters the batch
                    engine.release(staging)
                    live.discard(staging)
                    pending = None
                    continue
                free = next((i for i, s in enumerate(slots) if s is None), None)
                if free is None:
                    break  # prefilled and waiting for a slot: admission stalls, decoding goes on
                engine.move(staging, free)
                live.discard(staging)
                live.add(free)
                slots[free] = pending
                pending = None
            if compact:
                _close_holes(engine, slots, live)
            active = [i for i, s in enumerate(slots) if s is not None]
            if not active:
                last_completion = None  # idle time without an active decode request is not a fairness gap
                continue
            rows = _row_bucket(active[-1] + 1, batch_size)
            m.decode_bytes += int(engine.step_bytes(rows)) if hasattr(engine, "step_bytes") else 0
            t0 = clock()
            engine.decode(1, batch=rows)  # the occupied prefix of the slots; idle rows inside it produce nothing
            engine.synchronize()
            now = clock()
            m.decode_time += now - t0
            m.decode_step_ms.append((now - t0) * 1e3)
            if last_completion is not None:
                gaps_ms.append((now - last_completion) * 1e3)
            last_completion = now
            for i in active:
                s = slots[i]
                s["count"] += 1
                s["ctx"] += 1
                m.generated_tokens += 1
                m.decode_tokens += 1
            snapshot()
            for i in active:
                if slots[i]["count"] >= slots[i]["req"].max_new_tokens:
                    engine.release(i)
                    live.discard(i)
                    slots[i] = None
    finally:
        for slot in list(live):
            try:
                engine.release(slot)
            except RuntimeError:
                pass
    return _finish_metrics(engine, m, gaps_ms)


def _finish_metrics(engine, m: "ServingMetrics", gaps_ms: list) -> "ServingMetrics":
    stats = engine.stats() if hasattr(engine, "stats") else {}
    m.reused_page_allocations = int(stats.get("reused_page_allocations", 0))
    m.decode_step_count = len(m.decode_step_ms)
    m.decode_step_median_ms = median(m.decode_step_ms)
    m.decode_step_p95_ms = nearest_rank(m.decode_step_ms, 0.95)
    m.decode_step_max_ms = max(m.decode_step_ms, default=0.0)
    m.decode_gap_count = len(gaps_ms)
    m.decode_gap_median_ms = median(gaps_ms)
    m.decode_gap_p95_ms = nearest_rank(gaps_ms, 0.95)
    m.decode_gap_max_ms = max(gaps_ms, default=0.0)
    return m


def _serve_requests_packed(engine, requests, *, batch_size, prefill_step, prefill_budget, page_size, kv_bytes_per_page,
                           capacity_pages, clock, staging_slots, compact=True) -> "ServingMetrics":
    """The serving loop with several staging slots: every turn admits requests into the free staging slots, sends the next
    chunk of every staged prompt through ONE packed prefill pass (at most ``prefill_budget`` rows together, ``prefill_step``
    per prompt, 16 prompts), moves the prompts that finished into free decode slots (admission order), then runs one decode
    step over the occupied prefix.  Same counters as the one-at-a-time loop."""
    m = ServingMetrics()
    staging_slots = min(staging_slots, 16)
    slots: list[dict | None] = [None] * batch_size
    staged: list[dict] = []  # admission order; each holds its staging slot index
    free_staging = list(range(batch_size, batch_size + staging_slots))
    next_idx = 0
    live: set[int] = set()
    gaps_ms: list[float] = []
    last_completion: float | None = None

    def snapshot():
        states = [s for s in slots if s is not None] + staged
        m.peak_active_requests = max(m.peak_active_requests, len(states))
        pages = sum((s["ctx"] + page_size - 1) // page_size for s in states)
        waste = sum((-s["ctx"]) % page_size for s in states if s["ctx"] > 0)
        m.peak_live_pages = max(m.peak_live_pages, pages)
        m.peak_capacity_pages = max(m.peak_capacity_pages, capacity_pages)
        m.peak_kv_bytes = max(m.peak_kv_bytes, capacity_pages * kv_bytes_per_page)
        if waste > m.peak_tail_waste_slots:
            m.peak_tail_waste_slots = waste
            m.peak_tail_waste_live_slots = pages * page_size
            m.peak_tail_waste_bytes = waste * (kv_bytes_per_page // page_size if page_size else 0)
            m.peak_tail_waste_fraction = waste / (pages * page_size) if pages else 0.0

    try:
        while next_idx < len(requests) or staged or any(s is not None for s in slots):
            m.turns += 1
            while free_staging and next_idx < len(requests):
                st = free_staging.pop(0)
                engine.begin(st)
                live.add(st)
                staged.append({"req": requests[next_idx], "offset": 0, "count": 0, "ctx": 0, "staging": st})
                next_idx += 1
            budget = prefill_budget
            chunks = []
            for p in staged:
                tokens = p["req"].prompt_token_ids
                rem = len(tokens) - p["offset"]
                if rem <= 0 or budget <= 0:
                    continue
                n = min(prefill_step, rem, budget)
                chunks.append((p, tokens[p["offset"]:p["offset"] + n], p["offset"] + n >= len(tokens)))
                budget -= n
            if chunks:
                t0 = clock()
                engine.prefill_packed([(p["staging"], chunk, last) for p, chunk, last in chunks])
                engine.synchronize()
                m.prefill_time += clock() - t0
                m.prefill_chunks += len(chunks)
                for p, chunk, last in chunks:
                    p["offset"] += len(chunk)
                    p["ctx"] += len(chunk)
                    if last:
                        p["count"] = 1
                        m.generated_tokens += 1
                snapshot()
            for p in list(staged):  # admission order: the first prompt that finished takes the first free slot
                if p["offset"] < len(p["req"].prompt_token_ids):
                    continue
                if p["count"] >= p["req"].max_new_tokens:  # a one-token request never enters the batch
                    engine.release(p["staging"])
                elif (free := next((i for i, s in enumerate(slots) if s is None), None)) is not None:
                    engine.move(p["staging"], free)
                    live.add(free)
                    slots[free] = p
                else:
                    continue  # prefilled and waiting for a slot
                live.discard(p["staging"])
                free_staging.append(p["staging"])
                staged.remove(p)
            if compact:
                _close_holes(engine, slots, live)
            active = [i for i, s in enumerate(slots) if s is not None]
            if not active:
                last_completion = None
                continue
            rows = _row_bucket(active[-1] + 1, batch_size)
            m.decode_bytes += int(engine.step_bytes(rows)) if hasattr(engine, "step_bytes") else 0
            t0 = clock()
            engine.decode(1, batch=rows)
            engine.synchronize()
            now = clock()
            m.decode_time += now - t0
            m.decode_step_ms.append((now - t0) * 1e3)
            if last_completion is not None:
                gaps_ms.append((now - last_completion) * 1e3)
            last_completion = now
            for i in active:
                s = slots[i]
                s["count"] += 1
                s["ctx"] += 1
                m.generated_tokens += 1
                m.decode_tokens += 1
            snapshot()
            for i in active:
                if slots[i]["count"] >= slots[i]["req"].max_new_tokens:
                    engine.release(i)
                    live.discard(i)
                    slots[i] = None
    finally:
        for slot in list(live):
            try:
                engine.release(slot)
            except RuntimeError:
                pass
    return _finish_metrics(engine, m, gaps_ms)


def report_lines(num_seqs: int, prompt_tokens: int, total_time: float, m: ServingMetrics) -> list[str]:
    """The report exactly as the reference prints it (benches/bench.py:765-830; drivers regex these lines,
    benches/bench_course_progression.py:103-105)."""
    def div(a, b):
        return a / b if b else 0.0

    gen = m.generated_tokens
    return [
        f"Requests: {num_seqs}, Prompt tokens: {prompt_tokens}, Generated tokens: {gen}",
        f"Time: {total_time:.2f}s, Output throughput: {div(gen, total_time):.2f} tok/s",
        f"Total throughput (prompt+output): {div(prompt_tokens + gen, total_time):.2f} tok/s",
        f"Prefill throughput: {div(prompt_tokens, m.prefill_time):.2f} tok/s",
        f"Decode throughput: {div(m.decode_tokens, m.decode_time):.2f} tok/s",
        f"Request throughput: {div(num_seqs, total_time):.2f} req/s",
        f"Peak active requests: {m.peak_active_requests}",
        f"Peak KV bytes: {m.peak_kv_bytes}",
        f"Peak live KV pages: {m.peak_live_pages}",
        f"Peak KV capacity pages: {m.peak_capacity_pages}",
        f"Peak tail waste slots: {m.peak_tail_waste_slots}",
        f"Tail-waste snapshot live slots: {m.peak_tail_waste_live_slots}",
        f"Tail-waste snapshot bytes: {m.peak_tail_waste_bytes}",
        f"Tail-waste snapshot fraction: {m.peak_tail_waste_fraction:.6f}",
        f"Decode step latency ms (median/p95/max): {m.decode_step_median_ms:.3f}/{m.decode_step_p95_ms:.3f}/{m.decode_step_max_ms:.3f}",
        f"Decode completion gap ms (median/p95/max): {m.decode_gap_median_ms:.3f}/{m.decode_gap_p95_ms:.3f}/{m.decode_gap_max_ms:.3f}",
        f"Reused page allocations: {m.reused_page_allocations}",
        f"Page-pool growths: {m.storage_growths}",
        f"Pages copied during pool growth: {m.copied_pages_on_growth}",
        f"Dense KV bytes copied during growth: {m.dense_growth_copy_bytes}",
        f"Dense KV bytes copied into batch tensors: {m.dense_staging_copy_bytes}",
        f"Paged KV bytes copied during pool growth: {m.paged_growth_copy_bytes}",
    ]


class ScheduleOnlyEngine:
    """The engine's slot interface against a COST MODEL instead of a GPU: a virtual clock advances by
    ``prefill_ms_per_token`` per prompt token and ``decode_ms(rows)`` per batched step.  Enforces the slot protocol (a slot
    is begun once, moved into a free slot, released once) so that scheduler bugs surface without hardware."""

    def __init__(self, slots: int, prefill_ms_per_token: float = 0.055, decode_ms=lambda rows: 1.1 + 0.07 * rows):
        self.slots = [None] * slots  # context length per live slot
        self.now = 0.0
        self.prefill_ms_per_token = prefill_ms_per_token
        self.decode_ms = decode_ms
        self.page_allocations = 0

    def clock(self) -> float:
        return self.now

    def begin(self, slot):
        if self.slots[slot] is not None:
            raise RuntimeError("slot already holds a sequence")
        self.slots[slot] = 0

    def prefill(self, slot, tokens, chunk=None, want_logits=True):
        if self.slots[slot] is None:
            raise RuntimeError("slot holds no sequence")
        self.slots[slot] += len(tokens)
        self.now += len(tokens) * self.prefill_ms_per_token * 1e-3

    def prefill_packed(self, chunks):
        """One pass over several slots' chunks: rows cost the same, the fixed per-pass overhead is paid once."""
        if len({c[0] for c in chunks}) != len(chunks):
            raise RuntimeError("a slot appears twice in a packed prefill")
        for slot, tokens, _last in chunks:
            if self.slots[slot] is None:
                raise RuntimeError("slot holds no sequence")
            self.slots[slot] += len(tokens)
        self.now += sum(len(c[1]) for c in chunks) * self.prefill_ms_per_token * 1e-3

    def move(self, src, dst):
        if self.slots[src] is None or self.slots[dst] is not None:
            raise RuntimeError("bad move")
        self.slots[dst], self.slots[src] = self.slots[src], None

    def decode(self, steps, batch=None):
        for i in range(batch):
            if self.slots[i] is not None:
                self.slots[i] += steps
        self.now += steps * self.decode_ms(batch) * 1e-3

    def release(self, slot):
        if self.slots[slot] is None:
            raise RuntimeError("slot holds no sequence")
        self.slots[slot] = None

    def synchronize(self):
        pass

    def step_bytes(self, batch=None):
        """W + 147,456 B x the live contexts of the step (Qwen3-4B W4: W = 2,136,832,000 B)."""
        return 2_136_832_000 + 147_456 * sum(c for c in self.slots[:batch] if c is not None)

    def stats(self):
        return {"reused_page_allocations": 0, "pages_in_use": 0}
