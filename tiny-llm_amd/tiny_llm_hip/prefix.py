"""Prefix / KV reuse behind the reference's ``KvPrefixGenerator`` call surface (src/tiny_llm_ref/agent/branching.py:22-208):
render + prefill ONE checkpoint prefix, then steer any number of continuations from it without prefilling it again.

    gen = KvPrefixGenerator(engine, tokenizer, max_tokens=64)
    cp = gen.save_checkpoint(messages)          # prefix rendered WITHOUT the generation prompt, prefilled exactly once
    branch = gen.fork()                         # shares only the frozen prefix
    branch.restore_checkpoint(cp)
    text = branch(messages + [steering])        # the steered prompt must EXTEND the saved token prefix; greedy decode
    branch.reuse                                # PrefixReuse(reused_tokens, layer_offsets, avoided_prefill_tokens)

The reference keeps the prefix as a tuple of dense per-layer (K, V) arrays and hands every continuation fresh
``TinyKvFullCache`` objects pointing at them (branching.py:193-208).  Here the prefix lives in the fused engine's page pool:
``save_checkpoint`` prefills it into a FROZEN slot, every continuation is ``tl_engine_fork`` of that slot into the working slot --
full pages shared by reference count, the partial tail page copied (copy-on-write; include/tinyllm_engine.h) -- followed by the
prefill of the suffix and greedy decode through the captured graph; the working slot's pages go back to the pool afterwards.
Same names, argument meaning, return types and error texts as the reference class; ``model`` is a ``DecodeEngine`` with at least
two slots (or an object with an ``.engine`` of that kind).  Checkpoints are per engine: ``fork()`` shares the frozen slot.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any

__all__ = ["AgentError", "ModelCheckpoint", "PrefixReuse", "KvPrefixGenerator"]


class AgentError(ValueError):
    """A recoverable error that can be returned to the model (reference: agent/protocol.py:10-11)."""


@dataclass(frozen=True)
class ModelCheckpoint:
    """Snapshot needed to resume from a cached prefix (reference: agent/checkpoint.py:15-35, same validation)."""

    conversation_position: int
    response_index: int
    cached_token_ids: tuple[int, ...]
    layer_offsets: tuple[int, ...]

    def __post_init__(self) -> None:
        if any(type(v) is not int or v < 0 for v in (self.conversation_position, self.response_index)):
            raise AgentError("checkpoint positions must be non-negative integers")
        if any(type(t) is not int or t < 0 for t in self.cached_token_ids):
            raise AgentError("cached token ids must be non-negative integers")
        if not self.layer_offsets or any(type(o) is not int or o < 0 for o in self.layer_offsets):
            raise AgentError("checkpoint needs non-negative layer offsets")
        if any(o != len(self.cached_token_ids) for o in self.layer_offsets):
            raise AgentError("layer offsets must match the cached token prefix")


@dataclass(frozen=True)
class PrefixReuse:
    """Observable token and cache positions reused by one continuation (reference: agent/branching.py:22-28)."""

    reused_tokens: int
    layer_offsets: tuple[int, ...]
    avoided_prefill_tokens: int


class _Prefix:
    """The frozen prefix shared by a generator and its forks: engine slot, token ids, checkpoint."""

    def __init__(self) -> None:
        self.checkpoint: ModelCheckpoint | None = None
        self.tokens: tuple[int, ...] = ()


class KvPrefixGenerator:
    """Greedy generation on the fused decode engine from one frozen KV prefix."""

    PREFIX_SLOT = 1  # frozen; decode steps run over slots [0, 1), so the working slot is 0
    WORK_SLOT = 0

    def __init__(self, model: Any, tokenizer: Any, max_tokens: int, enable_thinking: bool = False, *, prefill_step: int = 2048) -> None:
        if type(max_tokens) is not int or max_tokens <= 0:
            raise ValueError("max_tokens must be a positive integer")
        if type(enable_thinking) is not bool:
            raise ValueError("enable_thinking must be a boolean")
        engine = getattr(model, "engine", model)
        layers = getattr(model, "num_hidden_layers", None) or getattr(engine, "num_hidden_layers", None)
        if type(layers) is not int or layers <= 0:
            raise ValueError("model must expose a positive num_hidden_layers")
        if getattr(engine, "max_batch", 0) < 2:
            raise ValueError("the engine needs two slots: one holds the frozen prefix, one the continuation")
        self._engine = engine
        self._model = model
        self._tokenizer = tokenizer
        self._max_tokens = max_tokens
        self._enable_thinking = enable_thinking
        self._layer_count = layers
        self._prefill_step = prefill_step
        self._response_index = 0
        self._prefix = _Prefix()
        self._restored = False
        self._reuse = PrefixReuse(0, (), 0)

    @property
    def reuse(self) -> PrefixReuse:
        """The prefix positions used by the latest continuation."""
        return self._reuse

    DECODE_BLOCK = 16  # decode steps per device round trip in __call__ (ids are read back and checked for EOS after each block)

    def save_checkpoint(self, messages: list) -> ModelCheckpoint:
        """Render and prefill one checkpoint prefix exactly once (into the frozen slot)."""
        if self._prefix.checkpoint is not None:
            raise AgentError("prefix checkpoint was already saved")
        token_ids = self._encode(self._render(messages, add_generation_prompt=False))
        if not token_ids:
            raise AgentError("checkpoint prompt must contain at least one token")
        self._engine.begin(self.PREFIX_SLOT)
        try:
            self._engine.prefill(self.PREFIX_SLOT, token_ids, chunk=self._prefill_step, want_logits=False)
            if self._engine.context_len(self.PREFIX_SLOT) != len(token_ids):
                raise AgentError("model did not populate every dense cache layer")
        except BaseException:  # a failed prefill must not leave the frozen slot live: a retry would find "slot is live"
            self._engine.release(self.PREFIX_SLOT)
            raise
        offsets = (len(token_ids),) * self._layer_count  # one sequence across all layers: every layer holds the whole prefix
        checkpoint = ModelCheckpoint(len(messages), self._response_index, token_ids, offsets)
        self._prefix.tokens = token_ids
        self._prefix.checkpoint = checkpoint
        self._reuse = PrefixReuse(len(token_ids), offsets, len(token_ids))
        return checkpoint

    def restore_checkpoint(self, checkpoint: ModelCheckpoint) -> None:
        """Bind a fresh continuation to this generator's frozen prefix."""
        if not isinstance(checkpoint, ModelCheckpoint):
            raise AgentError("model checkpoint is invalid")
        if self._prefix.checkpoint is None or checkpoint != self._prefix.checkpoint:
            raise AgentError("model checkpoint does not match the saved KV prefix")
        self._response_index = checkpoint.response_index
        self._restored = True

    def fork(self) -> "KvPrefixGenerator":
        """A fresh generator that shares only the immutable prefix (the frozen engine slot)."""
        if self._prefix.checkpoint is None:
            raise AgentError("save a prefix checkpoint before forking")
        branch = KvPrefixGenerator(self._model, self._tokenizer, self._max_tokens, self._enable_thinking, prefill_step=self._prefill_step)
        branch._prefix = self._prefix
        branch._response_index = self._prefix.checkpoint.response_index
        branch._reuse = self._reuse
        return branch

    def __call__(self, messages: list) -> str:
        cp = self._prefix.checkpoint
        if not self._restored or cp is None:
            raise AgentError("restore the checkpoint before generating")
        token_ids = self._encode(self._render(messages, add_generation_prompt=True))
        prefix_size = len(self._prefix.tokens)
        if token_ids[:prefix_size] != self._prefix.tokens:
            raise AgentError("steered prompt does not extend the saved token prefix")
        suffix = token_ids[prefix_size:]
        if not suffix:
            raise AgentError("steered prompt must add tokens after the saved prefix")
        eng = self._engine
        eng.fork(self.PREFIX_SLOT, self.WORK_SLOT)  # shared full pages, own tail page: the prefix is NOT prefilled again
        eos = self._tokenizer.eos_token_id
        output: list[int] = []
        try:
            eng.prefill(self.WORK_SLOT, suffix, chunk=self._prefill_step)  # its last row yields the first generated token
            # Decode in blocks and stop at the end-of-sequence id, as the reference's loop does (agent/branching.py:139-150): a
            # continuation that ends early neither pays for max_tokens steps nor runs into the page / context limit behind it.
            produced = eng.read_tokens(self.WORK_SLOT, 1)
            done = produced[0] == eos or self._max_tokens <= 0  # the reference's `for _ in range(max_tokens)` emits nothing at max_tokens <= 0
            if not done:
                output.append(int(produced[0]))
            while not done and len(output) < self._max_tokens:
                block = min(self.DECODE_BLOCK, self._max_tokens - len(output))
                # a block reserves pages / context step by step and stops where the slot's room ends: the steps that did run count --
                # an end-of-sequence id among them ends the continuation cleanly, as in the reference's token-by-token loop; only a
                # continuation that is still going when the room ends is the caller's error
                before, refused = eng.context_len(self.WORK_SLOT), None
                try:
                    eng.decode(block, batch=1)
                    ran = block
                except RuntimeError as error:
                    ran, refused = eng.context_len(self.WORK_SLOT) - before, error
                for token in (eng.read_tokens(self.WORK_SLOT, ran) if ran > 0 else []):  # greedy ids; the reference stops BEFORE emitting the end-of-sequence id
                    if token == eos:
                        done = True
                        break
                    output.append(int(token))
                if refused is not None and not done and len(output) < self._max_tokens:
                    raise refused
        finally:
            eng.release(self.WORK_SLOT)
        self._response_index += 1
        self._reuse = PrefixReuse(prefix_size, cp.layer_offsets, prefix_size)
        return self._tokenizer.decode(output)

    def close(self) -> None:
        """Give the frozen slot's pages back (the reference's arrays are garbage-collected; engine pages are not)."""
        if self._prefix.checkpoint is not None:
            self._engine.release(self.PREFIX_SLOT)
            self._prefix.checkpoint = None
            self._prefix.tokens = ()

    # -- rendering / encoding: the reference's helpers (agent/branching.py:165-191), same error texts ---------------------------
    def _render(self, messages: list, *, add_generation_prompt: bool) -> str:
        try:
            prompt = self._tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=add_generation_prompt,
                                                         enable_thinking=self._enable_thinking)
        except (KeyError, TypeError, ValueError) as error:
            raise AgentError("could not render checkpoint messages") from error
        if not isinstance(prompt, str):
            raise AgentError("chat template must render text")
        return prompt

    def _encode(self, prompt: str) -> tuple[int, ...]:
        try:
            tokens = tuple(int(t) for t in self._tokenizer.encode(prompt, add_special_tokens=False))
        except (TypeError, ValueError) as error:
            raise AgentError("tokenizer returned invalid token ids") from error
        if any(t < 0 for t in tokens):
            raise AgentError("tokenizer returned invalid token ids")
        return tokens
