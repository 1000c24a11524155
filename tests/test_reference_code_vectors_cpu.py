"""CPU tier: tests/golden/reference_code_vectors.npz -- logits produced by the REFERENCE'S OWN PYTHON SOURCES (Week-1 model and
the Week-2 readable `kv-cache` checkpoint, /root/reference/src/tiny_llm_ref imported as it is; the arithmetic underneath is the
torch facade of mlx, see tests/golden/make_reference_code_vectors.py) -- against

  * the product's host mirror on host tensors: BIT-identical (the same statement tests/test_reference_differential_cpu.py
    makes live, here against the committed file, i.e. also where /root/reference does not exist);
  * the numpy oracle and the float64 truth: the reference's own bf16 pipeline and the oracle's restatement of it sit equally far
    from the truth (that distance is the E every tolerance of this repository is built from);
  * a fresh run of the generator, where /root/reference is present: the committed file is current.
"""

import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import TINY_CFG, to_mlx_shaped
from oracle import tiny_oracle as O

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden" / "reference_code_vectors.npz"
CASES = {  # the generator's table (tests/golden/make_reference_code_vectors.py)
    "tiny_p5": (dict(), 3), "tiny_p37": (dict(), 3), "tiny_p150": (dict(), 3),
    "untied_gqa3_p23": (dict(hidden_size=384, num_attention_heads=3, num_key_value_heads=1, intermediate_size=640, num_hidden_layers=3,
                             tie_word_embeddings=False), 12),
}


@pytest.fixture()
def one_thread():
    """The goldens were produced with one torch thread: bit equality needs the same summation order."""
    before = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(before)


def expect_bits(got: np.ndarray, want: np.ndarray, what: str) -> None:
    """Bit equality -- on a machine whose CPU / BLAS build orders the fp32 sums like the one that generated the file.  Elsewhere
    the last bit of a bf16 logit may differ: then at most two bf16 steps of the largest logit are accepted, with a warning that
    says so (the LIVE statement of bit equality, same process and same machine, is tests/test_reference_differential_cpu.py)."""
    if np.array_equal(got, want):
        return
    worst = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    allowed = 2 * 2.0 ** -7 * float(np.abs(want).max())
    assert worst <= allowed, f"{what}: max |difference| {worst:.4f} > {allowed:.4f}"
    import warnings

    warnings.warn(f"{what}: not bit-identical on this machine (max |difference| {worst:.4f}, {float(np.mean(got != want)) * 100:.1f}% of the "
                  "elements): the CPU / BLAS build sums in another order than the one that generated tests/golden/reference_code_vectors.npz")


def from_bits(a) -> np.ndarray:
    return (np.asarray(a, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_mirror_reproduces_the_reference_code_bit_for_bit(golden, name, one_thread):
    from tiny_llm_hip import Qwen3ModelWeek1, Qwen3ModelWeek2

    overrides, wseed = CASES[name]
    cfg = dict(TINY_CFG, **overrides)
    model = to_mlx_shaped(cfg, O.make_qwen3_weights(cfg, seed=wseed, sigma=0.05), device="cpu")
    prompt, ids = golden[f"{name}/prompt"].tolist(), golden[f"{name}/ids"].tolist()
    week2 = Qwen3ModelWeek2(model, checkpoint="kv-cache")
    cache = week2.create_kv_cache()
    logits = week2(torch.tensor([prompt], dtype=torch.int32), 0, cache)
    assert logits.dtype == torch.bfloat16
    expect_bits(logits[0, -8:].float().numpy(), from_bits(golden[f"{name}/week2_kv_cache_prefill_logits_last8"]), f"{name}: prefill rows")
    rows, offset = [logits[0, -1]], len(prompt)
    for tok in ids[:-1]:
        rows.append(week2(torch.tensor([[tok]], dtype=torch.int32), offset, cache, logits_to_keep=1)[0, -1])
        offset += 1
    for layer_cache in cache:
        layer_cache.release()
    got = torch.stack(rows).float().numpy()
    expect_bits(got, from_bits(golden[f"{name}/week2_kv_cache_step_logits"]), f"{name}: decode steps")
    if np.array_equal(got, from_bits(golden[f"{name}/week2_kv_cache_step_logits"])):
        assert [int(np.argmax(r)) for r in got] == ids
    if f"{name}/week1_logits_last8" in golden:
        week1 = Qwen3ModelWeek1(model)(torch.tensor([prompt], dtype=torch.int32))[0, -8:].float().numpy()
        expect_bits(week1, from_bits(golden[f"{name}/week1_logits_last8"]), f"{name}: Week-1 rows")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_is_as_close_to_the_truth_as_the_reference_code(golden, name):
    overrides, wseed = CASES[name]
    cfg = dict(TINY_CFG, **overrides)
    w = O.make_qwen3_weights(cfg, seed=wseed, sigma=0.05)
    prompt, ids = golden[f"{name}/prompt"].tolist(), golden[f"{name}/ids"].tolist()
    reference_rows = from_bits(golden[f"{name}/week2_kv_cache_step_logits"]).astype(np.float64)
    oracle, truth = O.OracleQwen3(cfg, w), O.TruthQwen3(cfg, w)
    o_rows, t_rows = [oracle.forward(prompt)[0, -1]], [truth.forward(prompt)[0, -1]]
    for tok in ids[:-1]:
        o_rows.append(oracle.forward([tok])[0, -1])
        t_rows.append(truth.forward([tok])[0, -1])
    o_rows, t_rows = np.stack(o_rows).astype(np.float64), np.stack(t_rows)
    e_reference = float(np.abs(reference_rows - t_rows).max())
    e_oracle = float(np.abs(o_rows - t_rows).max())
    print(f"{name}: max |reference code - truth| = {e_reference:.4f}, max |oracle - truth| = {e_oracle:.4f}, "
          f"max |oracle - reference code| = {float(np.abs(o_rows - reference_rows).max()):.4f}")
    ulp = 2.0 ** -7 * float(np.abs(t_rows).max())
    assert e_oracle <= 1.5 * e_reference + ulp and e_reference <= 1.5 * e_oracle + ulp
    assert float(np.abs(o_rows - reference_rows).max()) <= e_oracle + e_reference


def batch_case(golden):
    prompts = [golden[f"batch4/prompt{i}"].tolist() for i in range(4)]
    return prompts, golden["batch4/ids"], from_bits(golden["batch4/prefill_last_logits"]), from_bits(golden["batch4/step_logits"])


def test_host_mirror_reproduces_the_reference_code_in_a_batch(golden, one_thread):
    """Four requests prefilled one by one, then decoded together on a BatchingKvCache (readable path): bit for bit."""
    from tiny_llm_hip import BatchingKvCache, Qwen3ModelWeek2

    prompts, ids, first, steps = batch_case(golden)
    model = Qwen3ModelWeek2(to_mlx_shaped(TINY_CFG, O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05), device="cpu"), checkpoint="kv-cache")
    batch = [BatchingKvCache(max_active_requests=4, max_seq_len=128) for _ in range(model.num_hidden_layers)]
    for rid, prompt in enumerate(prompts):
        own = model.create_kv_cache()
        logits = model(torch.tensor([prompt], dtype=torch.int32), 0, own, logits_to_keep=1)
        expect_bits(logits[0, -1].float().numpy(), first[rid], f"batch: prefill of request {rid}")
        for layer_batch, layer_own in zip(batch, own):
            layer_batch.add_request(layer_own, rid)
    offsets = [len(p) for p in prompts]
    for step in range(steps.shape[0]):
        logits = model(torch.tensor(ids[step], dtype=torch.int32).reshape(-1, 1), torch.tensor(offsets, dtype=torch.int32), batch, logits_to_keep=1)
        expect_bits(logits[:, -1].float().numpy(), steps[step], f"batch: step {step}")
        offsets = [o + 1 for o in offsets]


def batch_truth_and_oracle(prompts, ids):
    """Teacher-forced on the reference's ids: per request the float64 truth and the bf16 oracle, [steps, 4, vocab] each."""
    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    t_rows, o_rows = [], []
    for r, prompt in enumerate(prompts):
        truth, oracle = O.TruthQwen3(TINY_CFG, w), O.OracleQwen3(TINY_CFG, w)
        truth.forward(prompt), oracle.forward(prompt)
        t_rows.append([truth.forward([int(ids[s, r])])[0, -1] for s in range(ids.shape[0] - 1)])
        o_rows.append([oracle.forward([int(ids[s, r])])[0, -1] for s in range(ids.shape[0] - 1)])
    return np.asarray(t_rows).transpose(1, 0, 2), np.asarray(o_rows, dtype=np.float64).transpose(1, 0, 2)


def test_oracle_is_as_close_to_the_truth_as_the_reference_code_in_a_batch(golden):
    prompts, ids, _, steps = batch_case(golden)
    truth, oracle = batch_truth_and_oracle(prompts, ids)
    e_reference, e_oracle = float(np.abs(steps - truth).max()), float(np.abs(oracle - truth).max())
    print(f"batch of 4: max |reference code - truth| = {e_reference:.4f}, max |oracle - truth| = {e_oracle:.4f}")
    ulp = 2.0 ** -7 * float(np.abs(truth).max())
    assert e_oracle <= 1.5 * e_reference + ulp and e_reference <= 1.5 * e_oracle + ulp


# ---- the reference's STACK: its Python sources on its own kernel code (tests/golden/make_reference_stack_vectors.py) ----------------
STACK = ROOT / "tests" / "golden" / "reference_stack_vectors.npz"
STACK_CASES = ("week3_paged_p20", "week2_kernels_p12")
STACK_CHUNK = 8


def stack_rows(model_forward, prompt, ids):
    """Feeds the prompt in chunks of STACK_CHUNK tokens, then the given ids one by one; the last-position logits of every call."""
    rows = []
    for start in range(0, len(prompt), STACK_CHUNK):
        rows.append(model_forward(prompt[start:start + STACK_CHUNK]))
    for tok in ids[:-1]:
        rows.append(model_forward([int(tok)]))
    return np.stack(rows)


@pytest.mark.parametrize("name", STACK_CASES)
def test_oracle_tracks_the_reference_stack(name):
    """Logits of the reference's own Python on the reference's own kernels (oracle/_ref) against the numpy oracle driven the same way:
    the oracle restates those kernels (bit-identical per kernel, tests/test_oracle_vs_reference_kernels_cpu.py), so through a whole
    model the two may drift apart only by single roundings -- at most two bf16 steps of the largest logit -- and sit equally far from
    the float64 truth."""
    stack = np.load(STACK)
    prompt, ids = stack[f"{name}/prompt"].tolist(), stack[f"{name}/ids"].tolist()
    reference_rows = from_bits(stack[f"{name}/logits"]).astype(np.float64)
    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    oracle, truth = O.OracleQwen3(TINY_CFG, w), O.TruthQwen3(TINY_CFG, w)
    o_rows = stack_rows(lambda toks: oracle.forward(toks)[0, -1], prompt, ids).astype(np.float64)
    t_rows = stack_rows(lambda toks: truth.forward(toks)[0, -1], prompt, ids)
    step = 2.0 ** -7 * float(np.abs(t_rows).max())
    e_reference, e_oracle = float(np.abs(reference_rows - t_rows).max()), float(np.abs(o_rows - t_rows).max())
    apart = float(np.abs(o_rows - reference_rows).max())
    print(f"{name}: max |reference stack - truth| = {e_reference:.4f}, max |oracle - truth| = {e_oracle:.4f}, max |oracle - reference stack| = "
          f"{apart:.4f} ({float(np.mean(o_rows == reference_rows)) * 100:.1f}% of the logits bit-identical)")
    assert apart <= 2 * step, (apart, step)
    assert e_oracle <= 1.5 * e_reference + step and e_reference <= 1.5 * e_oracle + step
    assert [int(np.argmax(r)) for r in reference_rows[len(reference_rows) - len(ids):]] == ids


@pytest.mark.skipif(not Path("/root/reference/src/tiny_llm_ref").is_dir(), reason="/root/reference is not present (GPU box)")
def test_committed_vectors_are_what_the_reference_code_produces_now(tmp_path):
    script = (ROOT / "tests" / "golden" / "make_reference_code_vectors.py").read_text().replace(
        'HERE / "reference_code_vectors.npz"', f'Path({str(tmp_path / "fresh.npz")!r})')
    (tmp_path / "make.py").write_text(script.replace("HERE = Path(__file__).resolve().parent", f"HERE = Path({str(GOLDEN.parent)!r})"))
    proc = subprocess.run([sys.executable, str(tmp_path / "make.py")], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    fresh, committed = np.load(tmp_path / "fresh.npz"), np.load(GOLDEN)
    assert sorted(fresh.files) == sorted(committed.files)
    for key in committed.files:
        if "logits" in key:
            expect_bits(from_bits(fresh[key]), from_bits(committed[key]), f"regenerated {key}")
        else:
            np.testing.assert_array_equal(fresh[key], committed[key], err_msg=key)
