"""CPU tier: the engine-backed ``KvPrefixGenerator`` (tiny_llm_hip/prefix.py; reference surface: src/tiny_llm_ref/agent/branching.py:22-208).

The engine needs a GPU; here an engine with DecodeEngine's slot API answers from the numpy oracle (fork = a copy of the source
slot's K/V state), so what is tested is the class: one prefill of the prefix, fork + suffix prefill + greedy decode per continuation,
EOS handling, reuse bookkeeping, and the reference's error behaviour.  Where /root/reference exists, the REFERENCE'S OWN class is run
beside it on the reference's own readable Week-2 model over the facade, on the same checkpoint, tokenizer and messages: same
checkpoints, same texts, same reuse records, same error messages.  On the device: tests/test_zz_prefix_generator_gpu.py."""

import copy
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import TINY_CFG, to_mlx_shaped
from oracle import tiny_oracle as O

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path("/root/reference")


class Tokenizer:
    """A deterministic stand-in: one token per character of the rendered text (ids 3 + ord % 60), '<eos>' = 2."""

    eos_token_id = 2

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False, enable_thinking=False):
        if not isinstance(messages, list):
            raise TypeError("messages must be a list")
        text = "".join(f"[{m['role']}]{m['content']}" for m in messages)
        return text + ("[assistant]" if add_generation_prompt else "")

    def encode(self, text, add_special_tokens=False):
        return [3 + (ord(ch) % 60) for ch in text]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


class OracleSlots:
    """DecodeEngine's slot API on OracleQwen3: begin / prefill / fork / decode / read_tokens / release / context_len."""

    def __init__(self, weights, max_batch=2):
        self.weights, self.max_batch, self.num_hidden_layers = weights, max_batch, TINY_CFG["num_hidden_layers"]
        self.slots, self.out, self.pending, self.prefilled = {}, {}, {}, []

    def begin(self, slot=0):
        assert slot not in self.slots
        self.slots[slot], self.out[slot] = O.OracleQwen3(TINY_CFG, self.weights), []

    def context_len(self, slot=0):
        return self.slots[slot].offset if slot in self.slots else 0

    def prefill(self, slot, tokens, *, chunk=2048, want_logits=True):
        logits = self.slots[slot].forward(list(tokens))[0, -1]
        self.prefilled.append((slot, len(tokens)))
        if want_logits:
            self.pending[slot] = int(np.argmax(logits))
            self.out[slot].append(self.pending[slot])

    def fork(self, src, dst):
        assert dst not in self.slots and src in self.slots
        self.slots[dst], self.out[dst] = copy.deepcopy(self.slots[src]), []

    def decode(self, steps, batch=None, use_graph=True):
        assert batch == 1
        for _ in range(steps):
            logits = self.slots[0].forward([self.pending[0]])[0, -1]
            self.pending[0] = int(np.argmax(logits))
            self.out[0].append(self.pending[0])

    def read_tokens(self, slot, count):
        return self.out[slot][-count:]

    def release(self, slot=0):
        del self.slots[slot]


MESSAGES = [{"role": "system", "content": "be brief"}, {"role": "user", "content": "count the files"}]
STEER_A = {"role": "user", "content": "use ls"}
STEER_B = {"role": "user", "content": "use find instead"}


@pytest.fixture()
def weights():
    return O.make_qwen3_weights(TINY_CFG, seed=5, sigma=0.05)


def plain_generation(weights, tok, messages, max_tokens):
    """No reuse: the whole steered prompt through a fresh model."""
    ids = tok.encode(tok.apply_chat_template(messages, add_generation_prompt=True))
    m = O.OracleQwen3(TINY_CFG, weights)
    logits = m.forward(ids)[0, -1]
    out = []
    for _ in range(max_tokens):
        t = int(np.argmax(logits))
        if t == tok.eos_token_id:
            break
        out.append(t)
        logits = m.forward([t])[0, -1]
    return tok.decode(out)


def test_continuations_reuse_the_prefix_and_match_generation_without_reuse(weights):
    from tiny_llm_hip.prefix import KvPrefixGenerator, PrefixReuse

    tok, eng = Tokenizer(), OracleSlots(weights)
    gen = KvPrefixGenerator(eng, tok, max_tokens=6)
    cp = gen.save_checkpoint(MESSAGES)
    n = len(cp.cached_token_ids)
    assert cp.layer_offsets == (n,) * TINY_CFG["num_hidden_layers"] and cp.conversation_position == 2 and cp.response_index == 0
    assert gen.reuse == PrefixReuse(n, cp.layer_offsets, n)
    texts = {}
    for name, steer in (("a", STEER_A), ("b", STEER_B)):
        branch = gen.fork()
        branch.restore_checkpoint(cp)
        texts[name] = branch(MESSAGES + [steer])
        assert branch.reuse == PrefixReuse(n, cp.layer_offsets, n)
        assert texts[name] == plain_generation(weights, tok, MESSAGES + [steer], 6)
    assert texts["a"] != texts["b"]
    # the prefix went through the model ONCE; every continuation prefilled only its suffix
    assert eng.prefilled[0] == (1, n) and all(slot == 0 and length < n for slot, length in eng.prefilled[1:])
    assert eng.context_len(1) == n and 0 not in eng.slots  # frozen slot untouched, working slot returned
    gen.close()
    assert not eng.slots


def test_error_behaviour_of_the_reference_class(weights):
    from tiny_llm_hip.prefix import AgentError, KvPrefixGenerator, ModelCheckpoint

    tok, eng = Tokenizer(), OracleSlots(weights)
    with pytest.raises(ValueError, match="max_tokens must be a positive integer"):
        KvPrefixGenerator(eng, tok, max_tokens=0)
    with pytest.raises(ValueError, match="enable_thinking must be a boolean"):
        KvPrefixGenerator(eng, tok, max_tokens=4, enable_thinking=1)
    gen = KvPrefixGenerator(eng, tok, max_tokens=4)
    with pytest.raises(AgentError, match="save a prefix checkpoint before forking"):
        gen.fork()
    with pytest.raises(AgentError, match="restore the checkpoint before generating"):
        gen(MESSAGES)
    with pytest.raises(AgentError, match="could not render checkpoint messages"):
        gen.save_checkpoint("not a list")
    cp = gen.save_checkpoint(MESSAGES)
    with pytest.raises(AgentError, match="prefix checkpoint was already saved"):
        gen.save_checkpoint(MESSAGES)
    with pytest.raises(AgentError, match="model checkpoint is invalid"):
        gen.restore_checkpoint("cp")
    other = ModelCheckpoint(cp.conversation_position, cp.response_index + 1, cp.cached_token_ids, cp.layer_offsets)
    with pytest.raises(AgentError, match="model checkpoint does not match the saved KV prefix"):
        gen.restore_checkpoint(other)
    gen.restore_checkpoint(cp)
    with pytest.raises(AgentError, match="steered prompt does not extend the saved token prefix"):
        gen([{"role": "user", "content": "something else"}])
    with pytest.raises(AgentError, match="layer offsets must match the cached token prefix"):
        ModelCheckpoint(1, 0, (1, 2, 3), (2, 2))


@pytest.mark.skipif(not (REFERENCE / "src" / "tiny_llm_ref" / "agent" / "branching.py").is_file(), reason="/root/reference is not present")
def test_side_by_side_with_the_reference_class(weights):
    """The reference's own KvPrefixGenerator on the reference's own Week-2 readable model (mlx = the torch facade, CPU) against the
    engine-backed class on the oracle engine: same checkpoint, same generated text for two steered continuations, same reuse
    records.  (Both models round at the reference's op boundaries; the TINY checkpoint's greedy ids have clear margins.)"""
    import subprocess

    code = r'''
import sys, json
sys.path[:0] = [%(compat)r, %(amd)r, %(ext)r, %(tests)r, %(root)r]
import importlib
import mlx.core as mx
sys.path.insert(0, %(refsrc)r)
for name in [k for k in sys.modules if k == "tiny_llm_ref" or k.startswith("tiny_llm_ref.")]:
    del sys.modules[name]
from tiny_llm_ref.agent.branching import KvPrefixGenerator as RefGen
from tiny_llm_ref.qwen3_week2 import Qwen3ModelWeek2
import numpy as np
from helpers import TINY_CFG, to_mlx_shaped
from oracle import tiny_oracle as O
from test_prefix_generator_cpu import Tokenizer, MESSAGES, STEER_A, STEER_B
w = O.make_qwen3_weights(TINY_CFG, seed=5, sigma=0.05)
with mx.stream(mx.cpu):
    model = Qwen3ModelWeek2(to_mlx_shaped(TINY_CFG, w, device="cpu"), checkpoint="kv-cache")
    gen = RefGen(model, Tokenizer(), 6)
    cp = gen.save_checkpoint(MESSAGES)
    out = {"cp": [cp.conversation_position, cp.response_index, list(cp.cached_token_ids), list(cp.layer_offsets)], "texts": [], "reuse": []}
    for steer in (STEER_A, STEER_B):
        b = gen.fork(); b.restore_checkpoint(cp)
        out["texts"].append(b(MESSAGES + [steer]))
        out["reuse"].append([b.reuse.reused_tokens, list(b.reuse.layer_offsets), b.reuse.avoided_prefill_tokens])
print("RESULT " + json.dumps(out))
''' % dict(compat=str(ROOT / "tiny-llm_amd" / "compat"), amd=str(ROOT / "tiny-llm_amd"), ext=str(ROOT / "tiny-llm_amd" / "extensions_hip"),
           tests=str(ROOT / "tests"), root=str(ROOT), refsrc=str(REFERENCE / "src"))
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    import json

    ref = json.loads([l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])

    from tiny_llm_hip.prefix import KvPrefixGenerator

    tok, eng = Tokenizer(), OracleSlots(weights)
    gen = KvPrefixGenerator(eng, tok, max_tokens=6)
    cp = gen.save_checkpoint(MESSAGES)
    assert [cp.conversation_position, cp.response_index, list(cp.cached_token_ids), list(cp.layer_offsets)] == ref["cp"]
    for i, steer in enumerate((STEER_A, STEER_B)):
        b = gen.fork()
        b.restore_checkpoint(cp)
        assert b(MESSAGES + [steer]) == ref["texts"][i]
        assert [b.reuse.reused_tokens, list(b.reuse.layer_offsets), b.reuse.avoided_prefill_tokens] == ref["reuse"][i]
