"""GPU tier (collected last): the default route wherever the qkv projection of a batched step runs on the K-sliced matmul (engine option "attn_qkv_partials" = 0 turns it
off) -- 5..64 rows of any model whose shapes the register-resident matmul of csrc/qmm6.h does not take, and of every model with that kernel switched
off (engine option "qmm6" = 0: the Qwen3-4B-shape cases below; since round 5 qmm6 takes qkv at every row count there and leaves no slices): the qkv projection's slice-reduction launch is
dropped and the decode-attention kernel adds the skinny matmul's fp32 slice partials itself (csrc/engine_kernels.h, QP; csrc/engine.hip
engine_linear `keep`).  The kernel adds the slices in the reduction kernel's order and rounds once like it, so the two routes must
agree BIT FOR BIT: same greedy tokens, same final logits, over several decode steps (the appended K/V rows feed later steps).
The default route is the one held against the oracle and the float64 truth elsewhere (tests/test_engine_gpu.py,
tests/test_engine_qwen4b_gpu.py); this file only has to show the opt-in route computes the same thing, and that it really
drops a launch per layer."""

import os

import numpy as np
import pytest
import torch

from helpers import QWEN4B_CFG, TINY_CFG, to_mlx_shaped
from oracle import tiny_oracle as O

# All cases have run on the device (round 3: profiles/r03_labs/opt_in_route_tests_first_run.log); the route is the default now,
# and option "attn_qkv_partials" = 0 is the route with the reduction launch it is compared with.
pytestmark = [pytest.mark.gpu]


def run(model, cfg, n_seq, steps, page_size, partials, profile=False, prompt_base=3, prompt_spread=19, no_qmm6=False):
    from tiny_llm_hip.engine import DecodeEngine

    rng = np.random.default_rng(500 + n_seq)
    prompts = [[int(t) for t in rng.integers(1, cfg["vocab_size"], size=prompt_base + (7 * i) % prompt_spread)] for i in range(n_seq)]
    # routes with a twin are engine options (tl_engine_set_option): the attention kernel adding the qkv slice planes itself (default since
    # round 3) against the reduction launch; qmm6 = 0 is the K-sliced route of rounds 2-3, where qkv leaves fp32 slice planes
    options = {"attn_qkv_partials": 1 if partials else 0}
    if no_qmm6:
        options["qmm6"] = 0
    eng = DecodeEngine(model, page_size=page_size, num_pages=n_seq * 3 + 2, max_batch=n_seq, max_prefill_rows=32, options=options)
    try:
        for i, p in enumerate(prompts):
            eng.begin(i)
            eng.prefill(i, p, chunk=32)
        first = eng.read_pending(n_seq)
        eng.decode(steps, batch=n_seq)
        logits = eng.logits(n_seq).clone()
        tokens = [eng.read_tokens(i, steps + 1) for i in range(n_seq)]
        prof = eng.profile_step(n_seq) if profile else None
        for i in range(n_seq):
            eng.release(i)
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()
    return first, tokens, logits, prof


@pytest.mark.parametrize("n_seq", [5, 8, 16, 33, 64])
def test_tiny_model_same_bits_with_and_without_the_reduction_launch(n_seq):
    """TINY_CFG: head_dim 128, 4 query heads on 2 KV heads -- a GQA group of TWO in a kernel that holds four query rows per
    workgroup (the clamped rows), 16-token pages so the window walk takes the per-row page ids."""
    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    model = to_mlx_shaped(TINY_CFG, w)
    a = run(model, TINY_CFG, n_seq, steps=6, page_size=16, partials=False)
    b = run(model, TINY_CFG, n_seq, steps=6, page_size=16, partials=True)
    assert a[0] == b[0] and a[1] == b[1], "greedy tokens differ"
    assert torch.equal(a[2].view(torch.int16), b[2].view(torch.int16)), "final logits differ in their bits"


@pytest.mark.parametrize("n_seq", [17, 24, 40])
def test_qwen3_4b_shapes_same_bits_and_one_launch_fewer_per_layer(n_seq):
    """Qwen3-4B's layer shapes (32 query heads on 8 KV heads, 2,560 wide: the qkv projection is cut into 4 slices), 3 layers,
    128-token pages (one scalar page id per stage)."""
    from tiny_llm_hip.synthetic import synthetic_qwen3

    cfg = dict(QWEN4B_CFG, num_hidden_layers=3)
    model = synthetic_qwen3(cfg, seed=4, sigma=0.02, device="cuda")
    a = run(model, cfg, n_seq, steps=4, page_size=128, partials=False, profile=True, no_qmm6=True)
    b = run(model, cfg, n_seq, steps=4, page_size=128, partials=True, profile=True, no_qmm6=True)
    assert a[0] == b[0] and a[1] == b[1], "greedy tokens differ"
    assert torch.equal(a[2].view(torch.int16), b[2].view(torch.int16)), "final logits differ in their bits"
    launches = [sum(v["launches"] for v in r[3]["kinds"].values()) for r in (a, b)]
    assert launches[1] == launches[0] - cfg["num_hidden_layers"], f"launches per step {launches}: expected one fewer per layer"


@pytest.mark.parametrize("n_seq", [17, 40])
def test_qwen3_4b_shapes_same_bits_on_the_matrix_core_walk(n_seq):
    """The same comparison where the window walk runs on the matrix cores (csrc/attn_mfma.h, QP: windows of 128 tokens and more --
    prompts of 130..319 tokens; the short prompts above stay on the 64-token windows of the VALU walk)."""
    from tiny_llm_hip.synthetic import synthetic_qwen3

    cfg = dict(QWEN4B_CFG, num_hidden_layers=3)
    model = synthetic_qwen3(cfg, seed=4, sigma=0.02, device="cuda")
    a = run(model, cfg, n_seq, steps=4, page_size=128, partials=False, prompt_base=130, prompt_spread=190, no_qmm6=True)
    b = run(model, cfg, n_seq, steps=4, page_size=128, partials=True, prompt_base=130, prompt_spread=190, no_qmm6=True)
    assert a[0] == b[0] and a[1] == b[1], "greedy tokens differ"
    assert torch.equal(a[2].view(torch.int16), b[2].view(torch.int16)), "final logits differ in their bits"
