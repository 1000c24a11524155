"""GPU tier (collected last): the default route of single-row decode (TL_WO_MERGES_ATTN=0 turns it off) -- single-row decode with the context split 2 / 4 / 8 ways: the merge launch
behind the decode-attention kernel is dropped and the wo GEMV forms the merged attention row from the split partials while it
stages it (csrc/qmv3.h PRO_ATTN_MERGE, csrc/engine.hip engine_wo_merge).  The staging repeats attn_merge_kernel's arithmetic term
for term, so the two routes must agree to the last bf16 step over several steps; the default route is the one held against the oracle
and the float64 truth elsewhere (tests/test_engine_qwen4b_gpu.py).  Qwen3-4B layer shapes: the merging GEMV is instantiated for
the plans a 4,096-wide wo takes (the tiny test model's 512-wide wo keeps the merge launch)."""

import os

import numpy as np
import pytest
import torch

from helpers import QWEN4B_CFG

# First device run in round 3 (profiles/r03_labs/opt_in_route_tests_first_run.log): 2 and 4 windows bit-identical to the merge launch,
# 8 windows one bf16 step apart in a few logits (the compiler contracts the two kernels' multiply-adds differently).  The route is
# the default now; TL_WO_MERGES_ATTN=0 is the route with the merge launch it is compared with.
pytestmark = [pytest.mark.gpu]

CFG = dict(QWEN4B_CFG, num_hidden_layers=3)


@pytest.fixture(scope="module")
def model():
    from tiny_llm_hip.synthetic import synthetic_qwen3

    return synthetic_qwen3(CFG, seed=4, sigma=0.02, device="cuda")


def run(model, prompt, steps, merging):
    from tiny_llm_hip.engine import DecodeEngine

    old = os.environ.pop("TL_WO_MERGES_ATTN", None)
    os.environ["TL_WO_MERGES_ATTN"] = "1" if merging else "0"  # read when the engine is created (default since round 3: 1)
    try:
        eng = DecodeEngine(model, page_size=128, num_pages=24, max_batch=1, max_prefill_rows=1024)
    finally:
        os.environ.pop("TL_WO_MERGES_ATTN", None)
        if old is not None:
            os.environ["TL_WO_MERGES_ATTN"] = old
    try:
        eng.begin(0)
        eng.prefill(0, prompt, chunk=1024)
        first = eng.read_pending(1)
        eng.decode(steps, batch=1)
        logits = eng.logits(1).clone()
        tokens = eng.read_tokens(0, steps + 1)
        prof = eng.profile_step(1)
        eng.release(0)
    finally:
        eng.close()
    return first, tokens, logits, prof


@pytest.mark.parametrize("prompt_len,splits", [(40, 1), (100, 2), (200, 4), (300, 4), (600, 4), (1200, 8), (2500, 16)])
def test_same_bits_with_and_without_the_merge_launch(model, prompt_len, splits):
    """Context buckets 64 / 128 / 256 tokens = 1 / 2 / 4 windows of 64 tokens, 512 / 1,024 = 4 windows of 128 / 256, 2,048 = 8 windows
    of 256, 4,096 = 16 (windows are a quarter of the bucket, 64 .. 256 tokens): the wo GEMV merges 2, 4 and 8 partials; one window
    has nothing to merge and 16 keep the column-parallel merge launch (both routes then run the same kernels)."""
    rng = np.random.default_rng(prompt_len)
    prompt = [int(t) for t in rng.integers(256, CFG["vocab_size"], size=prompt_len)]
    a = run(model, prompt, steps=5, merging=False)
    b = run(model, prompt, steps=5, merging=True)
    # Same arithmetic, term for term; the compiler is free to contract a multiply-add in one kernel and not in the other, so the
    # merged row may differ in the last bit of an element and a logit by a bf16 step of the largest logit (measured: 2 and 4
    # windows identical, 8 windows one step in a few logits; the greedy ids of these seeded cases are the same).
    la, lb = a[2].float(), b[2].float()
    step = 2.0 ** -7 * float(la.abs().max().clamp(min=1.0))
    assert float((la - lb).abs().max()) <= 2 * step, "final logits differ by more than two bf16 steps of the largest logit"
    assert a[0] == b[0] and a[1] == b[1], "greedy tokens differ"
    assert a[3]["n_splits"] == splits, f"the split plan changed: {a[3]['n_splits']} windows"
    merges = [r[3]["kinds"]["attention_merge"]["launches"] for r in (a, b)]
    assert merges[0] == (CFG["num_hidden_layers"] if splits > 1 else 0)
    assert merges[1] == (0 if splits in (2, 4, 8) else merges[0]), f"merge launches per step {merges}"
