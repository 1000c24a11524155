"""GPU tier (collected last): the default route of 1-4-row decode since round 3 (TL_GEMV_WEIGHTED_ROWS=0 turns it off) -- the wo GEMV
also leaves h * post_attention_layernorm (bf16) and the gate|up GEMV stages that row and multiplies its finished sums by the row's
1 / rms (csrc/qmv3.h PRO_RMS_WEIGHTED, csrc/engine.hip weighted_rows_apply).  RMSNorm is x * inv * w with ONE scalar inv per row,
so the projection is linear in it; what changes is the rounding point of the staged element (bf16(x * w) instead of the
reference's bf16(x * inv * w), qwen3 week 2: rms_norm then quantized_matmul), one bf16 rounding per element either way.  The two
routes are therefore NOT bit-identical; they must stay within a few bf16 steps of each other after several layers and steps, and
the default route is the one held against the oracle and the float64 truth elsewhere (tests/test_engine_qwen4b_gpu.py).
Qwen3-4B layer shapes (the route needs the MFMA GEMV plans of the real wo / gate|up).

Two checkpoints: on the FLAT one (sigma 0.02 everywhere: near-tied logits, a one-step difference re-seeds the sequence) one decode
step is compared, logits only; on the PEAKED one (tests/test_engine_qwen4b_gpu.py: top-2 margins above one logit) five steps, logits
and greedy ids.  First device run with five steps on the flat checkpoint: three cases within 2 steps, the four-row case 159 steps
apart after a greedy id flipped in one row (both routes right, different sequences) -- hence the split."""

import os

import numpy as np
import pytest
import torch

from helpers import QWEN4B_CFG

pytestmark = [pytest.mark.gpu]

CFG = dict(QWEN4B_CFG, num_hidden_layers=3)


@pytest.fixture(scope="module")
def model():
    from tiny_llm_hip.synthetic import synthetic_qwen3

    return synthetic_qwen3(CFG, seed=6, sigma=0.02, device="cuda")


@pytest.fixture(scope="module")
def peaked_model():
    from tiny_llm_hip.synthetic import synthetic_qwen3

    return synthetic_qwen3(CFG, seed=11, sigma=0.02, device="cuda", embed_sigma=0.25, residual_gain=0.02)


def run(model, prompts, steps, weighted):
    from tiny_llm_hip.engine import DecodeEngine

    old = os.environ.pop("TL_GEMV_WEIGHTED_ROWS", None)
    os.environ["TL_GEMV_WEIGHTED_ROWS"] = "1" if weighted else "0"  # read when the engine is created (default: 1)
    try:
        eng = DecodeEngine(model, page_size=128, num_pages=16, max_batch=len(prompts), max_prefill_rows=1024)
    finally:
        os.environ.pop("TL_GEMV_WEIGHTED_ROWS", None)
        if old is not None:
            os.environ["TL_GEMV_WEIGHTED_ROWS"] = old
    try:
        for slot, prompt in enumerate(prompts):
            eng.begin(slot)
            eng.prefill(slot, prompt, chunk=1024)
        eng.decode(steps, batch=len(prompts))
        logits = eng.logits(len(prompts)).clone()
        tokens = [eng.read_tokens(slot, steps + 1) for slot in range(len(prompts))]
        for slot in range(len(prompts)):
            eng.release(slot)
    finally:
        eng.close()
    return tokens, logits


CASES = [(1, 40), (1, 200), (2, 90), (4, 60)]


def prompts_for(rows, prompt_len):
    rng = np.random.default_rng(100 * rows + prompt_len)
    return [[int(t) for t in rng.integers(256, CFG["vocab_size"], size=prompt_len + 3 * i)] for i in range(rows)]


def steps_apart(a, b):
    la, lb = a[1].float(), b[1].float()
    assert torch.isfinite(lb).all()
    step = 2.0 ** -7 * float(la.abs().max().clamp(min=1.0))  # one bf16 step of the largest logit
    return float((la - lb).abs().max()) / step


@pytest.mark.parametrize("rows,prompt_len", CASES)
def test_one_step_on_the_flat_checkpoint_stays_within_a_few_bf16_steps(model, rows, prompt_len):
    """One row with 1 window (plain wo GEMV) and 4 windows (the wo GEMV that merges the attention partials), 2 and 4 rows (the
    two- and four-row GEMV plans): the same prefill, then ONE decode step through 3 layers on both routes."""
    prompts = prompts_for(rows, prompt_len)
    a = run(model, prompts, steps=1, weighted=False)
    b = run(model, prompts, steps=1, weighted=True)
    worst = steps_apart(a, b)
    assert worst <= 4, f"logits {worst:.1f} bf16 steps apart after one step"
    assert worst > 0, "the two routes gave the same bits: is the weighted route running?"


@pytest.mark.parametrize("rows,prompt_len", CASES)
def test_five_steps_on_the_peaked_checkpoint_same_ids_and_close_logits(peaked_model, rows, prompt_len):
    prompts = prompts_for(rows, prompt_len)
    a = run(peaked_model, prompts, steps=5, weighted=False)
    b = run(peaked_model, prompts, steps=5, weighted=True)
    assert a[0] == b[0], "greedy tokens differ"
    worst = steps_apart(a, b)
    assert worst <= 6, f"final logits {worst:.1f} bf16 steps apart"
