"""GPU tier (collected last: written after the last GPU run of round 2): more than eight prompts ending in ONE
packed prefill pass.  The lm_head of a packed pass runs over the last rows of all ending prompts; both GEMV kernels hold at
most 8 activation rows, so 9..16 rows must go through in passes of 8 (csrc/engine.hip engine_qmv) -- before that fix rows
8.. of such a pass were never computed.  Every row is held against the oracle and the float64 truth like a solo prefill's."""

import numpy as np
import pytest

from helpers import TINY_CFG, check_against_truth, to_mlx_shaped
from oracle import tiny_oracle as O

# First device run: profiles/r02_labs/zz_gpu_tests_first_device_run.log (all passed).
pytestmark = [pytest.mark.gpu]


def prompt_ids(n, seed=0):
    rng = np.random.default_rng(seed)
    return [int(t) for t in rng.integers(1, TINY_CFG["vocab_size"], size=n)]


@pytest.mark.parametrize("n_prompts", [9, 12, 16])
def test_packed_prefill_with_more_than_eight_ending_prompts(n_prompts):
    from tiny_llm_hip.engine import DecodeEngine

    w = O.make_qwen3_weights(TINY_CFG, seed=3, sigma=0.05)
    model = to_mlx_shaped(TINY_CFG, w)
    # every prompt longer than 8 tokens: packed rows always take the GEMM path, and so does the oracle for more than 8 rows
    prompts = [prompt_ids(9 + 2 * i, seed=900 + i) for i in range(n_prompts)]
    oracle_rows = [O.OracleQwen3(TINY_CFG, w).forward(p)[0, -1] for p in prompts]
    truth_rows = [O.TruthQwen3(TINY_CFG, w).forward(p)[0, -1] for p in prompts]
    eng = DecodeEngine(model, page_size=16, num_pages=96, max_batch=n_prompts, max_prefill_rows=512)
    try:
        for s in range(n_prompts):
            eng.begin(s)
        eng.prefill_packed([(s, prompts[s], True) for s in range(n_prompts)])
        got = eng.logits(n_prompts).float().cpu().numpy()
        # all rows in one statement (HIP error <= 1.5 x the oracle's own error over the same rows), and no single row far out:
        # a row the lm_head pass skipped would be garbage, tens of logit units away
        check_against_truth(got, np.stack(oracle_rows), np.stack(truth_rows), what=f"packed prefill of {n_prompts} ending prompts")
        worst_oracle = float(np.abs(np.stack(oracle_rows) - np.stack(truth_rows)).max())
        for s in range(n_prompts):
            assert float(np.abs(got[s] - truth_rows[s]).max()) <= 3.0 * worst_oracle + 0.05, f"row {s}"
        first = eng.read_pending(n_prompts)
        for s in range(n_prompts):  # the pending token of every slot is the greedy id of ITS row
            assert int(first[s]) == int(np.argmax(got[s])), f"slot {s}"
        for s in range(n_prompts):
            eng.release(s)
        assert eng.stats()["pages_in_use"] == 0
    finally:
        eng.close()
