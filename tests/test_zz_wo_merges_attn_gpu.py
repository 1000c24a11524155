"""GPU tier (collected last): TL_WO_MERGES_ATTN=1 -- single-row decode with the context split 2 / 4 / 8 ways: the merge launch
behind the decode-attention kernel is dropped and the wo GEMV forms the merged attention row from the split partials while it
stages it (csrc/qmv3.h PRO_ATTN_MERGE, csrc/engine.hip engine_wo_merge).  The staging repeats attn_merge_kernel's arithmetic term
for term, so the two routes must agree BIT FOR BIT over several steps; the default route is the one held against the oracle
and the float64 truth elsewhere (tests/test_engine_qwen4b_gpu.py).  Qwen3-4B layer shapes: the merging GEMV is instantiated for
the plans a 4,096-wide wo takes (the tiny test model's 512-wide wo keeps the merge launch)."""

import os

import numpy as np
import pytest
import torch

from helpers import QWEN4B_CFG

# Written after the round's GPU budget was spent, never run: a kernel nobody has rehearsed can do worse than fail (a memory fault
# ends the whole pytest process), so this file runs only when TL_UNREHEARSED_GPU_TESTS=1 (tools/gpu_call_p.sh sets it) -- remove the
# gate after that run.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("TL_UNREHEARSED_GPU_TESTS") != "1",
                                                  reason="never run on the device yet: TL_UNREHEARSED_GPU_TESTS=1 (tools/gpu_call_p.sh) runs it")]

CFG = dict(QWEN4B_CFG, num_hidden_layers=3)


@pytest.fixture(scope="module")
def model():
    from tiny_llm_hip.synthetic import synthetic_qwen3

    return synthetic_qwen3(CFG, seed=4, sigma=0.02, device="cuda")


def run(model, prompt, steps, merging):
    from tiny_llm_hip.engine import DecodeEngine

    old = os.environ.pop("TL_WO_MERGES_ATTN", None)
    if merging:
        os.environ["TL_WO_MERGES_ATTN"] = "1"  # read when the engine is created
    try:
        eng = DecodeEngine(model, page_size=128, num_pages=8, max_batch=1, max_prefill_rows=1024)
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


@pytest.mark.parametrize("prompt_len,splits", [(40, 1), (100, 2), (200, 4), (300, 8), (600, 16)])
def test_same_bits_with_and_without_the_merge_launch(model, prompt_len, splits):
    """Context buckets 64 / 128 / 256 / 512 / 1,024 tokens = 1 / 2 / 4 / 8 / 16 windows of 64 tokens: the wo GEMV merges 2, 4 and 8
    partials; one window has nothing to merge and 16 keep the column-parallel merge launch (both routes then run the same kernels)."""
    rng = np.random.default_rng(prompt_len)
    prompt = [int(t) for t in rng.integers(256, CFG["vocab_size"], size=prompt_len)]
    a = run(model, prompt, steps=5, merging=False)
    b = run(model, prompt, steps=5, merging=True)
    assert a[0] == b[0] and a[1] == b[1], "greedy tokens differ"
    assert torch.equal(a[2].view(torch.int16), b[2].view(torch.int16)), "final logits differ in their bits"
    assert a[3]["n_splits"] == splits, f"the split plan changed: {a[3]['n_splits']} windows"
    merges = [r[3]["kinds"]["attention_merge"]["launches"] for r in (a, b)]
    assert merges[0] == (CFG["num_hidden_layers"] if splits > 1 else 0)
    assert merges[1] == (0 if splits in (2, 4, 8) else merges[0]), f"merge launches per step {merges}"
