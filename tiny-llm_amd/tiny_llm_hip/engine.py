"""Fused decode engine: host mirror of include/tinyllm_engine.h.

The reference decodes by calling its operators one by one from Python (``Qwen3ModelWeek3.__call__``,
src/tiny_llm_ref/qwen3_week3.py:320-338, driven by benches/bench.py:run_one_request_week2 277-312).  That
loop is kept (``Qwen3ModelWeek2/3`` in this package run it op by op on the HIP operators); this module is
the production path behind it: the same arithmetic issued as 5 fused kernels per layer inside one replayed
hipGraph, with tokens / context lengths / block tables resident on the device.

``DecodeEngine.from_model(mlx_shaped_model)`` re-packs the checkpoint once (QKV rows concatenated, gate/up
rows interleaved) and then exposes a request-level API: ``begin`` / ``prefill`` / ``decode`` / ``release``.
"""

from __future__ import annotations

import ctypes
import os
from typing import Any, Sequence

import torch

from ._ext import tiny_llm_ext_hip as _ext

_lib = _ext.lib()


def _w4(weight: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor) -> "_ext.TlW4":
    rows, words = weight.shape
    return _ext.TlW4(weight.data_ptr(), scales.data_ptr(), biases.data_ptr(), rows, words * 8)


def _bits(t: torch.Tensor) -> torch.Tensor:
    return t.view(torch.int32) if t.dtype != torch.int32 else t


class _Fused:
    """Keeps the re-packed tensors alive for the lifetime of the engine."""

    def __init__(self, weight, scales, biases):
        self.weight = _bits(weight).contiguous()
        self.scales = scales.to(torch.bfloat16).contiguous()
        self.biases = biases.to(torch.bfloat16).contiguous()

    @staticmethod
    def concat(layers: Sequence[Any]) -> "_Fused":
        return _Fused(torch.cat([_bits(l.weight) for l in layers], 0), torch.cat([l.scales for l in layers], 0),
                      torch.cat([l.biases for l in layers], 0))

    @staticmethod
    def interleave(a: Any, b: Any) -> "_Fused":
        def il(x, y):
            return torch.stack([x, y], dim=1).reshape(x.shape[0] * 2, *x.shape[1:])

        return _Fused(il(_bits(a.weight), _bits(b.weight)), il(a.scales, b.scales), il(a.biases, b.biases))

    def c(self) -> "_ext.TlW4":
        return _w4(self.weight, self.scales, self.biases)


class DecodeEngine:
    """Owns the fused weights, the paged KV pools and the captured decode graphs for one model on one GPU."""

    def __init__(self, mlx_model: Any, *, page_size: int = 128, num_pages: int = 512, max_batch: int = 1,
                 max_pages_per_seq: int | None = None, max_prefill_rows: int = 2048, options: dict | None = None,
                 kv_format: str = "bf16"):
        """``kv_format``: "bf16" (the reference's cache) or "fp8" -- K / V pages as OCP FP8 E4M3 codes with one power-of-two scale per
        row (tl_engine_create_kv; head_dim 128; the reference has no quantised cache, README.md:134-135: an extension, off by default).
        ``options`` (tests / lab tools only): routes with an A/B twin, by the names of tl_engine_set_option (include/tinyllm_engine.h),
        e.g. {"qmm7": 0}; the environment variable TL_ENGINE_OPTIONS ("qmm7=0,aql_fences=1") adds to them for the A/B scripts under tools/."""
        args = mlx_model.args
        if not torch.cuda.is_available():
            raise RuntimeError("DecodeEngine: the course extension is GPU-only")
        self.args = args
        self._keep: list[Any] = []
        layers = (_ext.TlLayerWeights * args.num_hidden_layers)()
        moe_layers: dict[int, Any] = {}  # Qwen3-MoE layers (reference qwen3_week3.py:209-214, 258-272): attached after create
        for i, layer in enumerate(mlx_model.model.layers):
            attn, mlp = layer.self_attn, layer.mlp
            qkv = _Fused.concat([attn.q_proj, attn.k_proj, attn.v_proj])
            wo = _Fused(attn.o_proj.weight, attn.o_proj.scales, attn.o_proj.biases)
            norms = [layer.input_layernorm.weight, layer.post_attention_layernorm.weight, attn.q_norm.weight,
                     attn.k_norm.weight]
            norms = [n.to(torch.bfloat16).contiguous() for n in norms]
            if hasattr(mlp, "switch_mlp"):  # router + stacked experts: no dense MLP in this layer
                moe_layers[i] = mlp
                none = _ext.TlW4(None, None, None, 0, 0)
                self._keep += [qkv, wo, norms]
                layers[i] = _ext.TlLayerWeights(qkv.c(), wo.c(), none, none, norms[0].data_ptr(), norms[1].data_ptr(),
                                                norms[2].data_ptr(), norms[3].data_ptr())
                continue
            gu = _Fused.interleave(mlp.gate_proj, mlp.up_proj)
            down = _Fused(mlp.down_proj.weight, mlp.down_proj.scales, mlp.down_proj.biases)
            self._keep += [qkv, wo, gu, down, norms]
            layers[i] = _ext.TlLayerWeights(qkv.c(), wo.c(), gu.c(), down.c(), norms[0].data_ptr(),
                                            norms[1].data_ptr(), norms[2].data_ptr(), norms[3].data_ptr())
        emb = mlx_model.model.embed_tokens
        embed = _Fused(emb.weight, emb.scales, emb.biases)
        final_norm = mlx_model.model.norm.weight.to(torch.bfloat16).contiguous()
        self._keep += [embed, final_norm]
        head = None
        if not getattr(args, "tie_word_embeddings", True):
            hl = mlx_model.lm_head
            head = _Fused(hl.weight, hl.scales, hl.biases)
            self._keep.append(head)
        if max_pages_per_seq is None:
            max_pages_per_seq = num_pages
        cfg = _ext.TlEngineConfig(
            args.hidden_size, args.num_hidden_layers, args.num_attention_heads, args.num_key_value_heads,
            args.head_dim, args.intermediate_size, args.vocab_size, float(args.rope_theta), float(args.rms_norm_eps),
            page_size, num_pages, max_batch, max_pages_per_seq, max_prefill_rows)
        self.page_size, self.max_batch, self.vocab_size = page_size, max_batch, args.vocab_size
        self.max_prefill_rows = int(max_prefill_rows)
        self.num_hidden_layers = int(args.num_hidden_layers)
        self.device = emb.weight.device
        handle = ctypes.c_void_p()
        embed_c = embed.c()
        head_c = head.c() if head is not None else None
        if kv_format not in ("bf16", "fp8"):
            raise ValueError("DecodeEngine: kv_format must be 'bf16' or 'fp8'")
        self.kv_format = kv_format
        _ext.check(_lib.tl_engine_create_kv(
            ctypes.byref(cfg), layers, ctypes.byref(embed_c), final_norm.data_ptr(),
            ctypes.byref(head_c) if head_c is not None else None, None,
            _ext.KV_FP8_E4M3 if kv_format == "fp8" else _ext.KV_BF16, ctypes.byref(handle)))
        self._h = handle
        opts = dict(item.split("=", 1) for item in os.environ.get("TL_ENGINE_OPTIONS", "").replace("+", ",").split(",") if "=" in item)
        opts.update(options or {})
        for name, value in opts.items():
            _ext.check(_lib.tl_engine_set_option(self._h, str(name).strip().encode(), int(value)))
        for i, mlp in moe_layers.items():
            self._attach_moe(i, mlp, args)

    def _attach_moe(self, layer: int, mlp: Any, args: Any) -> None:
        """Hand the router and the stacked experts of one sparse layer to the engine (tl_engine_set_moe_layer)."""
        router = _Fused(mlp.gate.weight, mlp.gate.scales, mlp.gate.biases)
        sw = mlp.switch_mlp
        experts = {}
        for name in ("gate_proj", "up_proj", "down_proj"):
            p = getattr(sw, name)
            w = _bits(p.weight).contiguous()
            if w.dim() != 3:
                raise ValueError(f"layer {layer}: switch_mlp.{name}.weight must be [experts, rows, words]")
            experts[name] = (w, p.scales.to(torch.bfloat16).contiguous(), p.biases.to(torch.bfloat16).contiguous())
        n_experts, inter, _ = experts["gate_proj"][0].shape
        self._keep += [router, experts]
        g, u, d = experts["gate_proj"], experts["up_proj"], experts["down_proj"]
        w = _ext.TlMoeWeights(router.c(), g[0].data_ptr(), u[0].data_ptr(), d[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                              u[1].data_ptr(), u[2].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), int(n_experts),
                              int(args.num_experts_per_tok), int(inter), int(bool(getattr(args, "norm_topk_prob", False))))
        _ext.check(_lib.tl_engine_set_moe_layer(self._h, layer, ctypes.byref(w)))

    @classmethod
    def from_model(cls, mlx_model: Any, **kwargs) -> "DecodeEngine":
        return cls(mlx_model, **kwargs)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.tl_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- slots -------------------------------------------------------------------------------------
    def begin(self, slot: int = 0) -> None:
        _ext.check(_lib.tl_engine_begin(self._h, slot))

    def reserve(self, slot: int, total_tokens: int) -> None:
        _ext.check(_lib.tl_engine_reserve(self._h, slot, total_tokens))

    def release(self, slot: int = 0) -> None:
        _ext.check(_lib.tl_engine_release(self._h, slot))

    def rewind(self, slot: int, n: int) -> None:
        _ext.check(_lib.tl_engine_rewind(self._h, slot, n))

    def move(self, src: int, dst: int) -> None:
        _ext.check(_lib.tl_engine_move(self._h, src, dst))

    def fork(self, src: int, dst: int) -> None:
        """Make free slot ``dst`` a copy-on-write twin of ``src`` (shared prefix pages, own tail page)."""
        _ext.check(_lib.tl_engine_fork(self._h, src, dst))

    def read_pending(self, count: int | None = None) -> list[int]:
        """Pending (= most recently generated) token id of slots [0, count); synchronises."""
        count = count or self.max_batch
        out = (ctypes.c_int32 * count)()
        _ext.check(_lib.tl_engine_read_pending(self._h, count, out))
        return list(out)

    def context_len(self, slot: int = 0) -> int:
        return _lib.tl_engine_context_len(self._h, slot)

    def set_token(self, slot: int, token: int) -> None:
        _ext.check(_lib.tl_engine_set_token(self._h, slot, int(token)))

    # -- compute -----------------------------------------------------------------------------------
    def prefill(self, slot: int, tokens: Sequence[int], *, chunk: int | None = None, want_logits: bool = True) -> None:
        """Chunked prefill (reference Request.try_prefill, batch.py:48-76): all chunks append K/V, the last one
        also produces the first generated token."""
        tokens = [int(t) for t in tokens]
        if not tokens:
            raise ValueError("prefill needs at least one token")
        chunk = self.max_prefill_rows if chunk is None else chunk  # (the largest chunk the engine was built for: 4,096-token chunks prefill at 100k tokens/s, 2,048 at 84k)
        for start in range(0, len(tokens), chunk):
            part = tokens[start:start + chunk]
            arr = (ctypes.c_int32 * len(part))(*part)
            last = start + chunk >= len(tokens)
            _ext.check(_lib.tl_engine_prefill(self._h, slot, arr, len(part), int(last and want_logits)))

    def prefill_packed(self, chunks: Sequence[tuple[int, Sequence[int], bool]]) -> None:
        """One pass of the multi-token path over several slots' chunks: ``chunks`` = (slot, token ids, ends_prompt) for up to 16
        slots, together at most ``max_prefill_rows`` tokens (tl_engine_prefill_packed).  A chunk that ends its prompt produces
        the slot's first generated token, like the last chunk of ``prefill``."""
        if not 1 <= len(chunks) <= 16:
            raise ValueError("prefill_packed takes between 1 and 16 chunks")
        flat = [int(t) for _, toks, _ in chunks for t in toks]
        n = len(chunks)
        slots = (ctypes.c_int * n)(*[int(c[0]) for c in chunks])
        lens = (ctypes.c_int * n)(*[len(c[1]) for c in chunks])
        want = (ctypes.c_int * n)(*[int(bool(c[2])) for c in chunks])
        arr = (ctypes.c_int32 * len(flat))(*flat)
        _ext.check(_lib.tl_engine_prefill_packed(self._h, n, slots, arr, lens, want))

    def verify(self, slot: int, tokens: Sequence[int]) -> list[int]:
        """Speculative verification: append 1..8 tokens to the slot and return, for each of them, the greedy token that
        follows it (reference speculative_generate's target call with logits_to_keep = all rows).  Synchronises."""
        tokens = [int(t) for t in tokens]
        if not 1 <= len(tokens) <= 8:
            raise ValueError("verify takes between 1 and 8 tokens")
        arr = (ctypes.c_int32 * len(tokens))(*tokens)
        out = (ctypes.c_int32 * len(tokens))()
        _ext.check(_lib.tl_engine_verify(self._h, slot, arr, len(tokens), out))
        return list(out)

    def decode(self, steps: int, batch: int | None = None, use_graph: bool = True) -> None:
        """Enqueue ``steps`` greedy decode steps over slots [0, batch); does not synchronise."""
        _ext.check(_lib.tl_engine_decode(self._h, batch or self.max_batch, int(steps), int(use_graph)))

    def read_tokens(self, slot: int, count: int) -> list[int]:
        out = (ctypes.c_int32 * count)()
        _ext.check(_lib.tl_engine_read_tokens(self._h, slot, count, out))
        return list(out)

    def logits(self, rows: int = 1) -> torch.Tensor:
        """A copy of the most recent logits [rows, vocab] (bf16)."""
        out = torch.empty((rows, self.vocab_size), dtype=torch.bfloat16, device=self.device)
        torch.cuda.current_stream().synchronize()  # `out` is allocated on torch's stream
        _ext.check(_lib.tl_engine_copy_logits(self._h, out.data_ptr(), rows))
        self.synchronize()
        return out

    def synchronize(self) -> None:
        _ext.check(_lib.tl_engine_synchronize(self._h))

    def step_bytes(self, batch: int | None = None) -> int:
        return int(_lib.tl_engine_step_bytes(self._h, batch or self.max_batch))

    PROFILE_KINDS = ("gemv_qkv", "gemv_o", "gemv_gate_up", "gemv_down", "gemv_lm_head", "attention",
                     "attention_merge", "step_end")

    def profile_step(self, batch: int | None = None) -> dict:
        """One real decode step with in-kernel clock stamps (tl_engine_profile_step): per-kind kernel time."""
        p = _ext.TlStepProfile()
        _ext.check(_lib.tl_engine_profile_step(self._h, batch or self.max_batch, ctypes.byref(p)))
        out = {"span_us": p.span_us, "clock_khz": p.clock_khz, "n_splits": p.n_splits, "kinds": {}}
        for i, name in enumerate(self.PROFILE_KINDS):
            out["kinds"][name] = {"us": p.kernel_us[i], "launches": p.launches[i],
                                  "bytes": p.gemv_bytes[i] if i < 5 else 0.0}
        return out

    def check_step(self, batch: int | None = None) -> dict:
        """One real decode step with the written-once checker behind every launch (tl_engine_check_step; test aid of the AQL route)."""
        c = _ext.TlStepCheck()
        _ext.check(_lib.tl_engine_check_step(self._h, batch or self.max_batch, ctypes.byref(c)))
        return {name: getattr(c, name) for name, _ in c._fields_}

    def replay_route(self) -> str:
        """"aql" (captured steps replay as AQL packets on the engine's own HSA queue) or "hipgraph: <why>" (include/tinyllm_engine.h)."""
        return (_lib.tl_engine_replay_route(self._h) or b"").decode()

    def stats(self) -> dict:
        s = _ext.TlEngineStats()
        _ext.check(_lib.tl_engine_get_stats(self._h, ctypes.byref(s)))
        return {name: getattr(s, name) for name, _ in s._fields_}

    # -- convenience: one request, like benches/bench.py:run_one_request_week2 --------------------------
    def generate(self, prompt: Sequence[int], max_new_tokens: int, *, slot: int = 0, chunk: int | None = None) -> list[int]:
        self.begin(slot)
        try:
            self.prefill(slot, prompt, chunk=chunk)
            if max_new_tokens > 1:
                self.decode(max_new_tokens - 1, batch=slot + 1)
            return self.read_tokens(slot, max_new_tokens)
        finally:
            self.release(slot)


# decode row counts used by the scheduler: exact up to 4 rows (fused GEMV), then the row-block sizes of the skinny matmul
_DECODE_ROW_BUCKETS = (1, 2, 3, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256)


def batch_generate_ids(engine: DecodeEngine, prompts: Sequence[Sequence[int]], max_new_tokens: int | Sequence[int],
                       batch_size: int, prefill_step: int = 128, eos_token_id: int | None = None,
                       on_step=None) -> list[tuple[int, list[int]]]:
    """Continuous batching over engine slots with the reference scheduler's shape (batch_generate,
    src/tiny_llm_ref/batch.py:136-285; benches/bench.py:run_batch_requests_serving 351-572): every loop turn
    (a) admits one pending request and prefills ONE chunk of at most ``prefill_step`` tokens in the staging slot,
    (b) adopts it into the lowest free decode slot once its prefill is complete, (c) runs one batched decode step over
    the occupied prefix of the slots (the reference steps all ``batch_size`` rows; idle rows carry context 0 and produce
    nothing, so skipping the idle tail changes no result), (d) retires finished requests and returns their pages.
    Token-id in, token-id out (no tokenizer can be downloaded here).  Needs ``engine.max_batch >= batch_size + 1``:
    the last slot is the prefill staging slot.  Returns [(prompt_idx, generated ids)] in completion order."""
    if batch_size <= 0 or prefill_step <= 0:
        raise ValueError("batch_size and prefill_step must be positive")
    if engine.max_batch < batch_size + 1:
        raise ValueError("engine needs batch_size + 1 slots (one prefill staging slot)")
    limits = [max_new_tokens] * len(prompts) if isinstance(max_new_tokens, int) else list(max_new_tokens)
    staging = batch_size
    queue = list(range(len(prompts)))
    slots: list[dict | None] = [None] * batch_size
    pending: dict | None = None
    finished: list[tuple[int, list[int]]] = []
    live_slots: set[int] = set()
    try:
        while queue or pending is not None or any(s is not None for s in slots):
            if queue and pending is None:
                idx = queue.pop(0)
                engine.begin(staging)
                live_slots.add(staging)
                pending = {"idx": idx, "tokens": [int(t) for t in prompts[idx]], "offset": 0, "out": [], "limit": limits[idx]}
            if pending is not None:
                total = len(pending["tokens"])
                if pending["offset"] < total:
                    chunk = pending["tokens"][pending["offset"]:pending["offset"] + prefill_step]
                    last = pending["offset"] + len(chunk) >= total
                    engine.prefill(staging, chunk, chunk=len(chunk), want_logits=last)
                    pending["offset"] += len(chunk)
                    if last:
                        pending["out"].append(engine.read_tokens(staging, 1)[0])
                if pending["offset"] >= total:
                    done = len(pending["out"]) >= pending["limit"] or pending["out"][-1] == eos_token_id
                    if done:
                        engine.release(staging)
                        live_slots.discard(staging)
                        finished.append((pending["idx"], pending["out"]))
                        pending = None
                    else:
                        free = next((i for i, s in enumerate(slots) if s is None), None)
                        if free is not None:
                            engine.move(staging, free)
                            live_slots.discard(staging)
                            live_slots.add(free)
                            slots[free] = pending
                            pending = None
            if any(s is not None for s in slots):
                # slots fill lowest-first, so rows above the highest occupied one are idle: decode only a bucket that
                # covers the occupied prefix (the engine keeps one captured graph per row count; buckets bound their number)
                def bucket(n):
                    return min(next((b for b in _DECODE_ROW_BUCKETS if b >= n), batch_size), batch_size)

                top = max(i for i, s in enumerate(slots) if s is not None) + 1
                count = sum(s is not None for s in slots)
                if bucket(count) < bucket(top):
                    # finished requests left holes below the highest live slot and closing them lowers the row bucket (a step
                    # costs by its rows): hand the highest live slots over to the holes -- block-table row, context length and
                    # pending token move, no K/V bytes (tl_engine_move); slot numbers are invisible to the results
                    lo, hi = 0, batch_size - 1
                    while True:
                        while lo < hi and slots[lo] is not None:
                            lo += 1
                        while hi > lo and slots[hi] is None:
                            hi -= 1
                        if lo >= hi:
                            break
                        engine.move(hi, lo)
                        slots[lo], slots[hi] = slots[hi], None
                        live_slots.discard(hi)
                        live_slots.add(lo)
                    top = count
                rows = bucket(top)
                engine.decode(1, batch=rows)
                tokens = engine.read_pending(rows)
                if on_step is not None:
                    on_step(sum(s is not None for s in slots))
                for i, req in enumerate(slots):
                    if req is None:
                        continue
                    req["out"].append(tokens[i])
                    if len(req["out"]) >= req["limit"] or tokens[i] == eos_token_id:
                        engine.release(i)
                        live_slots.discard(i)
                        finished.append((req["idx"], req["out"]))
                        slots[i] = None
    finally:
        for slot in list(live_slots):
            try:
                engine.release(slot)
            except RuntimeError:
                pass
    return finished


def speculative_generate_ids(target: DecodeEngine, draft: DecodeEngine, prompt: Sequence[int], max_new_tokens: int,
                             proposal_length: int = 4, eos_token_id: int | None = None, *, slot: int = 0,
                             chunk: int = 2048, stats: dict | None = None) -> list[int]:
    """Greedy speculative decoding over two engines (reference speculative_generate, src/tiny_llm_ref/generate.py:84-322;
    token ids in, token ids out).  The draft engine free-runs ``proposal_length`` fused decode steps on the device; the
    target scores the pending token plus the proposals in one ``verify`` call (at most 8 rows through the paged decode
    kernel) and both KV caches are rewound to the accepted prefix.  Returns exactly the target's greedy continuation
    (up to ``max_new_tokens`` ids, stopping before ``eos_token_id``); ``stats`` receives call and acceptance counts."""
    if not isinstance(proposal_length, int) or isinstance(proposal_length, bool) or proposal_length < 0:
        raise ValueError("proposal_length must be a non-negative integer")
    if proposal_length > 7:
        raise ValueError("proposal_length must be at most 7 (8 verification rows)")
    prompt = [int(t) for t in prompt]
    if not prompt:
        raise ValueError("prompt must hold at least one token")
    counts = {"target_calls": 0, "draft_steps": 0, "proposed": 0, "accepted": 0}
    out: list[int] = []
    target.begin(slot)
    draft_live = False
    try:
        target.prefill(slot, prompt, chunk=chunk)
        token = target.read_tokens(slot, 1)[0]
        counts["target_calls"] += 1
        if proposal_length > 0:
            draft.begin(slot)
            draft_live = True
            draft.prefill(slot, prompt, chunk=chunk, want_logits=False)
        while len(out) < max_new_tokens and token != eos_token_id:
            room = max_new_tokens - len(out) - 1          # proposals that could still be emitted after `token`
            k = min(proposal_length, room)
            proposals: list[int] = []
            if k > 0:
                draft.set_token(slot, token)
                draft.decode(k, batch=slot + 1)
                proposals = draft.read_tokens(slot, k)
                counts["draft_steps"] += k
                if eos_token_id in proposals:              # the draft stops proposing after its own EOS
                    keep = proposals.index(eos_token_id) + 1
                    draft.rewind(slot, k - keep)
                    proposals = proposals[:keep]
            fed = [token] + proposals
            predicted = target.verify(slot, fed)
            counts["target_calls"] += 1
            counts["proposed"] += len(proposals)
            own = [token] + predicted[:-1]                 # what the target alone would have fed at each row
            cut = next((i for i, (mine, given) in enumerate(zip(own, fed))
                        if mine != given or mine == eos_token_id), None)
            if cut is None:                                # everything accepted: the last prediction is a bonus token
                out.extend(own)
                counts["accepted"] += len(proposals)
                if proposals:                              # the draft has not consumed its last proposal yet
                    draft.decode(1, batch=slot + 1)
                    counts["draft_steps"] += 1
                token = predicted[-1]
                continue
            out.extend(own[:cut])
            counts["accepted"] += cut - 1
            target.rewind(slot, len(fed) - cut)
            if proposals:
                draft.rewind(slot, len(proposals) - cut)
            token = own[cut]
        if stats is not None:
            stats.update(counts)
        return out[:max_new_tokens]
    finally:
        target.release(slot)
        if draft_live:
            draft.release(slot)
